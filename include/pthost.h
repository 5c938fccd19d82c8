/*
 * pthost.h — C entry points of the host scene layer (libpthost.so).
 *
 * The host layer restates the reference's scene-side code in C++ (pathtracer_amd/host/):
 * primitive constructors, Instance transforms, PrimitiveBuffer::build_bvh with the binned
 * SAH builder of src/bvh.rs:289-516, Camera::new of src/camera.rs:79-144.  These entry
 * points let a non-C++ caller (the Python tests and benchmark here; a Rust `-sys` crate
 * in the reference) hand over an authoring-level description and get back the flat,
 * built arrays that libptcore consumes.
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "ptcore.h"
#include "ptscene_src.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* Builds primitives, instance matrices, BVHs (where a set asks for them) and camera
 * constants from the description.  Returns NULL and fills err on failure — including the
 * case where the reference would panic (BVH deeper than 63 levels, src/bvh.rs:484-486). */
pth_scene *pth_scene_build(const pts_scene *scene, const pts_camera *camera, char *err, size_t errlen);
/* The same, leaving triangle buffers of at least skip_tri_bvh_from triangles (0: none) without a BVH, in
 * insertion order, for a device library that builds its own (PT_OPT_DEVICE_BVH of ptcore.h). */
pth_scene *pth_scene_build_opts(const pts_scene *scene, const pts_camera *camera, uint32_t skip_tri_bvh_from, char *err, size_t errlen);
void pth_scene_free(pth_scene *s);

/* Flat views; valid until pth_scene_free. */
const pt_scene_desc *pth_scene_desc(const pth_scene *s);
const pt_camera_desc *pth_camera_desc(const pth_scene *s);

/* perm[i] = insertion index (within its set and type) of the primitive now at position i
 * of buffer `type` (PT_PRIM_*) of set 0 (objects) or 1 (lights). Returns the length. */
uint32_t pth_perm(const pth_scene *s, int set, int type, uint32_t *out);

/* Camera::render(objects, lights, buffer): uploads the scene to `ctx`, renders every
 * stratum of the camera, writes W*H*3 bytes to rgb8.  Returns a PT_ERR_* code. */
int pth_camera_render(const pth_scene *s, pt_context *ctx, uint64_t seed, uint8_t *rgb8);

/* ---- Wavefront .obj files (TriMesh::load_as_vec, src/objects/tri_mesh.rs:299-381).
 * One mesh per model of the file, in file order, with tobj's default options (separate
 * position / normal / texcoord index streams, faces not triangulated).  NULL on failure
 * (the reference logs and continues with no meshes); `err` receives the reason. */
typedef struct pth_obj pth_obj;
typedef struct pth_obj_mesh {
    const float *positions;           /* [n_positions][3], in order of first use by the model's faces */
    uint32_t n_positions;
    const uint32_t *indices;          /* [n_indices] into positions */
    uint32_t n_indices;
    const float *normals;             /* [n_normals][3] or NULL: the model has no normals */
    uint32_t n_normals;
    const uint32_t *normal_indices;   /* [n_indices] into normals */
    const float *texcoords;           /* [n_texcoords][2] or NULL */
    uint32_t n_texcoords;
    const uint32_t *texcoord_indices; /* [n_indices] into texcoords */
} pth_obj_mesh;
pth_obj *pth_obj_load(const char *path, char *err, size_t errlen);
void pth_obj_free(pth_obj *o);
uint32_t pth_obj_mesh_count(const pth_obj *o);
int pth_obj_mesh_get(const pth_obj *o, uint32_t mesh, pth_obj_mesh *out); /* views valid until pth_obj_free */
/* TriMesh::to_triangles (tri_mesh.rs:383-415): index streams read three at a time.  Returns
 * the triangle count (-1: bad mesh index or an index stream that is not a multiple of three)
 * and fills pos [n][3][3], nrm [n][3][3], uv [n][3][2] where the pointer is not NULL and the
 * mesh has that attribute; has_normals / has_uvs report which it has. */
int64_t pth_obj_triangles(const pth_obj *o, uint32_t mesh, float *pos, float *nrm, float *uv, int *has_normals, int *has_uvs);

/* Image files.  Replaces image::open(path)?.to_rgb8() of Image::load (src/textures/image.rs:17-26): a PNG file
 * (8- or 16-bit grey, grey + alpha, RGB, RGBA or palette, not interlaced) or a PPM file (P6 / P3) as RGB8 texels,
 * row-major, top row first: the layout pts_texture.texels takes.  Call once with rgb8 = NULL for the size, then
 * with a buffer of width * height * 3 bytes (capacity = its size).  Returns 0, or 1 with the reason in err; the
 * reference answers a failed load with its default checker texture, which a pts_texture of type PTS_TEX_IMAGE
 * with no texels gives. */
int pth_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb8, size_t capacity, char *err, size_t errlen);
/* Replaces RgbImage::save("strat.png") (src/main.rs:93): an 8-bit RGB PNG of the bytes pth_camera_render or
 * pt_read_image returned.  Returns 0, or 1 when the file cannot be written. */
int pth_image_save_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb8);

#ifdef __cplusplus
}
#endif
#endif /* PTHOST_H */
