/*
 * ptscene_src.h — authoring-level scene description (plain C structs, no pointers owned).
 *
 * This is the scene exactly as a user of the reference renderer writes it in
 * the src/scenes files, before anything is derived from it: primitives in insertion
 * order, instance transforms as a list of operations, materials and textures by
 * their constructor arguments, and the arguments of Camera::new.  Every derived
 * quantity (quad plane constants, instance matrices and inverses, camera basis,
 * bounding boxes, BVH permutation and node arrays) is computed by whoever
 * consumes this description, so that two independent consumers can be compared.
 *
 * Reference constructors restated as data:
 *   Triangle::new            src/objects/tri_mesh.rs:29-75
 *   Sphere::new              src/objects/sphere.rs:19-27
 *   Quad::new                src/objects/quad_mesh.rs:25-52
 *   Instance::{rotate*,translate*,scale*}   src/objects.rs:306-380
 *   PrimitiveBuffer / LightBuffer           src/objects.rs:130-233
 *   materials                src/materials/{lambertian,metal,dielectric,light,disney}.rs
 *   textures                 src/textures.rs:49-104, src/textures/image.rs:6-29
 *   Camera::new              src/camera.rs:79-90
 */
#ifndef PTSCENE_SRC_H
#define PTSCENE_SRC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- textures (src/textures.rs) ---------------------------------------- */
enum {
    PTS_TEX_SOLID = 0,     /* Solid::new(albedo)                       color[] */
    PTS_TEX_CHECKERED = 1, /* Checkered::new(odd, even, inv_scale)     odd, even, scale */
    PTS_TEX_IMAGE = 2,     /* Image::new(Some(img)) / Image::load(..)  image (or -1 = load failed) */
    PTS_TEX_NOISE = 3      /* not in the reference: color * (1 + sin(scale * p.z + 10 * turbulence(p))) with
                              gradient noise on the integer lattice, gradients hashed from `odd` (a seed)
                              and the lattice point, 7 octaves; color[], scale, odd */
};

typedef struct pts_texture {
    uint32_t type;
    float color[3];
    int32_t odd;   /* texture index */
    int32_t even;  /* texture index */
    float scale;   /* "inv_scale" field of Checkered */
    int32_t image; /* index into images[], -1 = None */
} pts_texture;

typedef struct pts_image {
    uint32_t width, height;
    uint64_t offset; /* byte offset of RGB8 texels (row-major, top row first) in texels[] */
} pts_image;

/* ---- materials (src/materials) ------------------------------------- */
enum {
    PTS_MAT_LAMBERTIAN = 0,       /* texture                                              */
    PTS_MAT_METAL = 1,            /* p[0..2]=albedo, p[3]=fuzz                             */
    PTS_MAT_DIELECTRIC = 2,       /* p[0]=ref_idx                                          */
    PTS_MAT_DIFFUSE_LIGHT = 3,    /* light::Diffuse: texture                               */
    PTS_MAT_DISNEY = 4,           /* texture; p[0..11] = subsurface, metallic,
                                     specular_transmission, specular, specular_tint,
                                     roughness, anisotropic, sheen, sheen_tint, clearcoat,
                                     clearcoat_gloss, ior  (Disney::new argument order)    */
    PTS_MAT_DISNEY_DIFFUSE = 5,   /* texture; p[0]=roughness, p[1]=subsurface              */
    PTS_MAT_DISNEY_METAL = 6,     /* texture; p[0]=roughness, p[1]=anisotropic             */
    PTS_MAT_DISNEY_GLASS = 7,     /* texture; p[0]=roughness, p[1]=anisotropic, p[2]=eta   */
    PTS_MAT_DISNEY_CLEARCOAT = 8, /* p[0]=clearcoat, p[1]=clearcoat_gloss                  */
    PTS_MAT_DISNEY_SHEEN = 9,     /* texture; p[0]=sheen, p[1]=sheen_tint                  */
    PTS_MAT_COUNT = 10
};

typedef struct pts_material {
    uint32_t type;
    int32_t texture; /* texture index, -1 where the type has none */
    float p[14];
} pts_material;

/* ---- primitive sets (PrimitiveBuffer, src/objects.rs:130-206) ----------- */
typedef struct pts_range {
    uint32_t first, count;
} pts_range;

typedef struct pts_primset {
    pts_range tris, spheres, quads, instances; /* ranges into the scene-wide arrays */
    uint32_t build_bvh;                        /* 1 = PrimitiveBuffer::build_bvh() was called */
    uint32_t reserved;
} pts_primset;

/* ---- instance transform operations (src/objects.rs:306-380) ------------- */
enum {
    PTS_OP_ROTATE = 0,      /* a[0]=angle, a[1..3]=axis      */
    PTS_OP_ROTATE_X = 1,    /* a[0]=angle                    */
    PTS_OP_ROTATE_Y = 2,
    PTS_OP_ROTATE_Z = 3,
    PTS_OP_TRANSLATE = 4,   /* a[0..2]=shift                 */
    PTS_OP_TRANSLATE_X = 5, /* a[0]                          */
    PTS_OP_TRANSLATE_Y = 6,
    PTS_OP_TRANSLATE_Z = 7,
    PTS_OP_SCALE = 8,       /* a[0..2]                       */
    PTS_OP_SCALE_X = 9,     /* a[0]                          */
    PTS_OP_SCALE_Y = 10,
    PTS_OP_SCALE_Z = 11,
    PTS_OP_SCALE_UNIFORM = 12 /* a[0]                        */
};

typedef struct pts_xform_op {
    uint32_t op;
    float a[4];
} pts_xform_op;

typedef struct pts_instance {
    uint32_t child_set; /* index into sets[] (Instance::from_obj(obj)) */
    uint32_t op_first, op_count;
    uint32_t reserved;
} pts_instance;

/* ---- camera (arguments of Camera::new, src/camera.rs:79-90) ------------- */
typedef struct pts_camera {
    float aspect_ratio;
    uint32_t image_width;
    float vfov;
    float look_from[3];
    float look_at[3];
    float focal_length;
    float defocus_angle;
    uint32_t spp;
    uint32_t max_depth;
    float background[3];
} pts_camera;

/* ---- participating media (not in the reference) --------------------------- */
/* A homogeneous medium filling a sphere, the "constant medium" of the Next Week scene.  A ray that
   crosses the sphere between its t_min and its closest surface hit scatters after a free flight of
   -ln(u) / density (u uniform in [0,1)) if that is shorter than its way through the sphere; the path
   continues from there in a uniformly drawn direction with its throughput multiplied by the albedo
   texture's value at that point.  Media are tried in index order, each against the closest event so
   far.  Media are not primitives: they are in no BVH and never appear in hit records. */
typedef struct pts_medium {
    float center[3];
    float radius;
    float density;
    int32_t texture; /* albedo */
} pts_medium;
enum { PTS_MAX_MEDIA = 8 };

/* ---- moving spheres (not in the reference) ---------------------------------- */
/* Motion blur of the Next Week scene: a sphere whose centre moves linearly from center0 at time 0 to
   center1 at time 1.  Every path has one time in [0,1), a hash of its generator's key that consumes none of
   its draws, and sees the sphere where it is then: Sphere::hit (src/objects/sphere.rs:56-99) with that centre.
   Moving spheres are tested after everything else (PrimitiveBuffer::hit, src/objects.rs:191-197), in index
   order, against the closest hit so far; they are in no set, no BVH and no light list. */
typedef struct pts_moving_sphere {
    float center0[3];
    float center1[3];
    float radius;
    int32_t material;
} pts_moving_sphere;
enum { PTS_MAX_MOVING = 8 };

/* ---- the scene ----------------------------------------------------------- */
enum { PTS_TRI_HAS_NORMALS = 1, PTS_TRI_HAS_UVS = 2 };

typedef struct pts_scene {
    uint32_t n_textures;
    const pts_texture *textures;
    uint32_t n_images;
    const pts_image *images;
    uint64_t n_texel_bytes;
    const uint8_t *texels;

    uint32_t n_materials;
    const pts_material *materials;

    /* Triangle::new(vertices, normals, texture_uv, material) per triangle. */
    uint32_t n_tris;
    const float *tri_pos;     /* [n][3 vertices][xyz]                               */
    const float *tri_nrm;     /* [n][3][xyz]   or NULL (no triangle has normals)    */
    const float *tri_uv;      /* [n][3][uv]    or NULL                              */
    const uint8_t *tri_flags; /* [n] PTS_TRI_HAS_* or NULL (= flags implied by non-NULL arrays) */
    const int32_t *tri_mat;   /* [n] material index, -1 = None                      */

    uint32_t n_spheres;
    const float *spheres; /* [n][cx,cy,cz,r] */
    const int32_t *sphere_mat;

    uint32_t n_quads;
    const float *quads; /* [n][origin xyz, u xyz, v xyz] */
    const int32_t *quad_mat;

    uint32_t n_instances;
    const pts_instance *instances;
    uint32_t n_ops;
    const pts_xform_op *ops;

    /* sets[0] = objects handed to Camera::render, sets[1] = LightBuffer.objects,
       sets[2..] = children of instances. */
    uint32_t n_sets;
    const pts_primset *sets;

    uint32_t n_media; /* at most PTS_MAX_MEDIA */
    const pts_medium *media;

    uint32_t n_moving; /* at most PTS_MAX_MOVING */
    const pts_moving_sphere *moving;
} pts_scene;

#ifdef __cplusplus
}
#endif
#endif /* PTSCENE_SRC_H */
