// Host-side scene layer of the MI355X path tracer (C++17).
//
// It keeps the reference renderer's scene-description API — the same type and method
// names with the same meaning — so that scene code written against the reference reads
// the same here, and turns a finished scene into the flat arrays of include/ptcore.h:
//
//   reference (Rust)                               here
//   ---------------------------------------------  ------------------------------------
//   TextureRef = Arc<dyn Texture>                  pt::TextureRef  (Solid, Checkered, Image)
//   MaterialRef = Arc<dyn Material>                pt::MaterialRef (Lambertian, Metal, Dielectric,
//                                                  Diffuse, Disney, DisneyDiffuse, DisneyMetal,
//                                                  DisneyGlass, DisneyClearcoat, DisneySheen)
//   Triangle / Sphere / Quad / Instance            pt::Triangle / Sphere / Quad / Instance
//   PrimitiveBuffer, LightBuffer                   pt::PrimitiveBuffer, pt::LightBuffer
//   BVHBuilder::build(SplitMethod::SAH, ..)        pt::BVHBuilder::build
//   Camera::new / Camera::render                   pt::Camera::Camera / pt::Camera::render
//
// Nothing here traces rays: Camera::render hands the scene to libptcore and fails if the
// device library reports an error.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ptcore.h"

namespace pt {

// ---- small f32 vector/matrix types, operations in nalgebra's evaluation order --------
struct Vector3 {
    float x = 0, y = 0, z = 0;
    Vector3() = default;
    Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
using Point3 = Vector3;
using Color = Vector3;
Vector3 operator+(Vector3 a, Vector3 b);
Vector3 operator-(Vector3 a, Vector3 b);
Vector3 operator-(Vector3 a);
Vector3 operator*(float s, Vector3 a);
Vector3 operator*(Vector3 a, float s);
Vector3 operator/(Vector3 a, float s);
float dot(Vector3 a, Vector3 b);
Vector3 cross(Vector3 a, Vector3 b);
Vector3 normalize(Vector3 a);

struct Matrix4 {
    float m[4][4];
    static Matrix4 identity();
    static Matrix4 new_rotation(Vector3 axisangle);
    static Matrix4 new_translation(Vector3 t);
    static Matrix4 new_nonuniform_scaling(Vector3 s);
    static Matrix4 new_scaling(float s);
    Vector3 transform_point(Point3 p) const;
};
Matrix4 operator*(const Matrix4 &a, const Matrix4 &b);

// ---- src/bvh.rs:33-281 ------------------------------------------------------------------
struct BBox {
    Point3 min, max;
    static BBox from_corners(Point3 a, Point3 b);  // BBox::new
    static BBox empty();
    static BBox point(Point3 p);
    Point3 centroid() const;
    BBox merge(const BBox &o) const;
    BBox enclose(Point3 p) const;
    float get_surface_area() const;
    int longest_axis() const;
    Vector3 offset(Point3 p) const;
    BBox transform(const Matrix4 &m) const;
};

// ---- textures (src/textures.rs, src/textures/image.rs) --------------------------------
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;  // RGB8, row-major, top row first
    RgbImage() = default;
    RgbImage(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 3, 0) {}
    bool save_ppm(const std::string &path) const;
    bool save_png(const std::string &path) const;  // 8-bit RGB, as RgbImage::save("*.png") of src/main.rs:93
    // PNG or PPM file -> RGB8 as image::open(path)?.to_rgb8() (src/textures/image.rs:18-19); null (and the reason in *why) on failure
    static std::shared_ptr<const RgbImage> load(const std::string &path, std::string *why = nullptr);
};
struct Texture;
using TextureRef = std::shared_ptr<const Texture>;
struct Texture {
    uint32_t type = PTS_TEX_SOLID;
    Color albedo;
    TextureRef odd, even;
    float inv_scale = 0;
    std::shared_ptr<const RgbImage> image;  // null = Image::load failed
    uint32_t seed = 0;                      // Noise
};
namespace Solid { TextureRef make(Color albedo); }
namespace Checkered {
TextureRef make(TextureRef odd, TextureRef even, float inv_scale);
TextureRef new_solid(Color odd, Color even, float inv_scale);
}
namespace Image {
TextureRef make(std::shared_ptr<const RgbImage> image);
TextureRef load(const std::string &path);  // src/textures/image.rs:17-26; a file that cannot be read gives the default texture
}
// Not in the reference: marble-like noise, PTS_TEX_NOISE of ptscene_src.h.
namespace Noise { TextureRef make(Color color, float scale, uint32_t seed); }

// ---- materials (src/materials/*.rs) ------------------------------------------------------
struct Material {
    uint32_t type = PTS_MAT_LAMBERTIAN;
    TextureRef texture;
    float p[14] = {0};
    float derived[8] = {0};
    bool is_emissive() const { return type == PTS_MAT_DIFFUSE_LIGHT; }
    bool is_specular() const { return type == PTS_MAT_METAL || type == PTS_MAT_DIELECTRIC || type == PTS_MAT_DIFFUSE_LIGHT; }
};
using MaterialRef = std::shared_ptr<const Material>;
namespace Lambertian {
MaterialRef make(Color albedo);                // Lambertian::new
MaterialRef new_texture(TextureRef texture);   // None -> gray(0.5)
}
namespace Metal { MaterialRef make(Color albedo, float fuzz); }
namespace Dielectric { MaterialRef make(float ref_idx); }
namespace Diffuse {  // the emitter of src/materials/light.rs
MaterialRef make(Color albedo);
MaterialRef gray(float v);
MaterialRef new_texture(TextureRef texture);
}
namespace Disney {
MaterialRef make(TextureRef base_color, float subsurface, float metallic, float specular_transmission, float specular,
                 float specular_tint, float roughness, float anisotropic, float sheen, float sheen_tint, float clearcoat,
                 float clearcoat_gloss, float ior);
}
namespace DisneyDiffuse { MaterialRef make(Color base_color, float roughness, float subsurface); MaterialRef new_from_texture(TextureRef t, float roughness, float subsurface); }
namespace DisneyMetal { MaterialRef make(Color base_color, float roughness, float anisotropic); MaterialRef new_from_texture(TextureRef t, float roughness, float anisotropic); }
namespace DisneyGlass { MaterialRef make(Color base_color, float roughness, float anisotropic, float eta); MaterialRef new_from_texture(TextureRef t, float roughness, float anisotropic, float eta); }
namespace DisneyClearcoat { MaterialRef make(float clearcoat, float clearcoat_gloss); }
namespace DisneySheen { MaterialRef make(Color base_color, float sheen, float sheen_tint); MaterialRef new_from_texture(TextureRef t, float sheen, float sheen_tint); }

// MaterialRegistry (src/materials.rs:125-158): name -> material while authoring a scene.
class MaterialRegistry {
public:
    void add_material(const std::string &name, MaterialRef m) { materials_[name] = std::move(m); }
    void create_material(const std::string &name, MaterialRef m) { add_material(name, std::move(m)); }
    MaterialRef get(const std::string &name) const;  // null (and a log line) when missing
private:
    std::unordered_map<std::string, MaterialRef> materials_;
};

// ---- primitives ---------------------------------------------------------------------------
struct Triangle {  // src/objects/tri_mesh.rs:21-75
    float data[8][3];
    MaterialRef material;
    Triangle(const std::array<Point3, 3> &vertices, const std::array<Vector3, 3> *normals, const std::array<std::array<float, 2>, 3> *texture_uv,
             MaterialRef material);
    Point3 position(int vertex) const { return Point3(data[0][vertex], data[1][vertex], data[2][vertex]); }
    BBox bbox() const;
};
// One model of a Wavefront .obj file, as tobj's default (multi-index, untriangulated) export
// gives it: src/objects/tri_mesh.rs:280-460.
struct TriMesh {
    std::vector<Point3> positions;
    std::vector<uint32_t> position_indices;
    bool has_normals = false;  // normals: Some(..) in the reference
    std::vector<Vector3> normals;
    std::vector<uint32_t> normal_indices;
    bool has_uvs = false;
    std::vector<std::array<float, 2>> uvs;
    std::vector<uint32_t> uv_indices;
    // One mesh per model, in file order; empty when the file cannot be loaded (the reference
    // logs and carries on with no meshes).  `error`, if given, receives the reason.
    static std::vector<TriMesh> load_as_vec(const std::string &path, std::string *error = nullptr);
    std::vector<Triangle> to_triangles() const;  // material None
    std::vector<Triangle> to_triangles_with_mat(MaterialRef material) const;
};
struct Sphere {  // src/objects/sphere.rs:9-27
    Point3 center;
    float radius;
    MaterialRef mat;
    BBox bbox_;
    Sphere(Point3 center, float radius, MaterialRef mat);
    BBox bbox() const { return bbox_; }
};
struct PrimitiveBuffer;
struct Quad {  // src/objects/quad_mesh.rs:10-102
    Point3 origin;
    Vector3 u, v, normal, w;
    float d, area;
    MaterialRef mat;
    BBox bbox_;
    Quad(Point3 origin, Vector3 u, Vector3 v, MaterialRef mat);
    static PrimitiveBuffer new_box(Point3 a, Point3 b, MaterialRef mat);
    BBox bbox() const { return bbox_; }
};
struct Instance {  // src/objects.rs:281-381
    std::shared_ptr<const PrimitiveBuffer> obj;
    Matrix4 transform, inverse;
    // Set by the scene flattening only, for an instance that sat inside other instances: `transform` / `inverse`
    // above are then the outermost level's, and these are the (transform, inverse) of the levels further in, outer to inner.
    std::vector<std::pair<Matrix4, Matrix4>> inner_levels;
    static Instance from_obj(PrimitiveBuffer obj);
    void rotate(float angle, Vector3 axis);
    void rotate_x(float angle);
    void rotate_y(float angle);
    void rotate_z(float angle);
    void translate(Vector3 shift);
    void translate_x(float x);
    void translate_y(float y);
    void translate_z(float z);
    void scale(Vector3 s);
    void scale_x(float x);
    void scale_y(float y);
    void scale_z(float z);
    void scale_uniform(float s);
    BBox bbox() const;
};

template <class T>
struct InnerPrimitiveBuffer {  // src/objects.rs:105-128
    std::vector<T> buffer;
    std::vector<pt_node> bvh;  // empty = None
    BBox bbox = BBox::empty();
    std::vector<uint32_t> perm;  // perm[i] = insertion index of buffer[i]
    void push(T primitive) {
        bbox = bbox.merge(primitive.bbox());
        perm.push_back((uint32_t)buffer.size());
        buffer.push_back(std::move(primitive));
    }
    size_t len() const { return buffer.size(); }
};

struct PrimitiveBuffer {  // src/objects.rs:130-206
    InnerPrimitiveBuffer<Triangle> triangles;
    InnerPrimitiveBuffer<Sphere> spheres;
    InnerPrimitiveBuffer<Quad> quads;
    InnerPrimitiveBuffer<Instance> instances;
    BBox bbox = BBox::empty();
    void add_triangle(Triangle t);
    void add_sphere(Sphere s);
    void add_quad(Quad q);
    void add_instance(Instance i);
    // Builds one BVH per non-empty typed buffer. Throws std::runtime_error where the reference
    // panics (a tree deeper than 63 levels, src/bvh.rs:484-486 + src/objects.rs:172).
    // skip_tris_from: triangle buffers with at least this many triangles are left without a BVH, for a
    // device library that builds its own (0: every buffer gets one, as in src/objects.rs:165-189)
    void build_bvh(size_t skip_tris_from = 0);
    size_t len() const { return triangles.len() + spheres.len() + quads.len() + instances.len(); }
};

struct LightBuffer {  // src/objects.rs:208-233
    PrimitiveBuffer objects;
    void add_triangle(Triangle t) { objects.add_triangle(std::move(t)); }
    void add_sphere(Sphere s) { objects.add_sphere(std::move(s)); }
    void add_quad(Quad q) { objects.add_quad(std::move(q)); }
    void add_instance(Instance i) { objects.add_instance(std::move(i)); }
};

// src/bvh.rs:289-516 with SplitMethod::SAH, 12 buckets, max leaf size 3.
// `order` starts as the identity over boxes and is permuted exactly as the reference
// permutes its primitive Vec; nodes come out in the reference's pre-order.
struct BVHBuilder {
    static bool build(const std::vector<BBox> &boxes, std::vector<uint32_t> &order, std::vector<pt_node> &nodes);
    template <class T>
    static bool build(InnerPrimitiveBuffer<T> &objects);
};

// ---- flattening ------------------------------------------------------------------------------
// Owns the arrays a pt_scene_desc points into.
class FlatScene {
public:
    // pre-registers materials/textures so that their indices follow a given order (optional)
    int material_index(const MaterialRef &m);
    int texture_index(const TextureRef &t);
    void build(const PrimitiveBuffer &objects, const LightBuffer &lights);
    const pt_scene_desc &desc() const { return desc_; }
    // Participating media (pts_medium; not in the reference).  Call before build().
    void add_medium(Point3 center, float radius, float density, const TextureRef &albedo);
    // Moving spheres (pts_moving_sphere; not in the reference).  Call before build().
    void add_moving_sphere(Point3 center0, Point3 center1, float radius, const MaterialRef &material);

private:
    uint32_t add_set(const PrimitiveBuffer &pb, const InnerPrimitiveBuffer<Instance> *instances);
    template <class T>
    pt_buffer_desc add_nodes(const InnerPrimitiveBuffer<T> &b, uint32_t prim_first);
    std::vector<float> tri_data_, spheres_, quads_;
    std::vector<int32_t> tri_mat_, sphere_mat_, quad_mat_;
    std::vector<pt_instance_desc> instances_, chain_;
    std::vector<pt_node> nodes_;
    std::vector<pt_primset_desc> sets_;
    std::vector<pt_material_desc> materials_;
    std::vector<pts_texture> textures_;
    std::vector<pts_image> images_;
    std::vector<uint8_t> texels_;
    std::vector<pts_medium> media_;
    std::vector<pts_moving_sphere> moving_;
    std::vector<float> moving_delta_;
    std::unordered_map<const Material *, int> mat_ids_;
    std::unordered_map<const Texture *, int> tex_ids_;
    std::unordered_map<const RgbImage *, int> img_ids_;
    std::vector<MaterialRef> keep_m_;
    std::vector<TextureRef> keep_t_;
    pt_scene_desc desc_{};
};

// ---- src/camera.rs ------------------------------------------------------------------------------
class Camera {
public:
    Camera(float aspect_ratio, uint32_t image_width, float vfov, Point3 look_from, Point3 look_at, float focal_length,
           float defocus_angle, uint32_t spp, uint32_t max_depth, Color background);
    uint32_t get_height() const { return desc_.image_height; }
    uint32_t get_width() const { return desc_.image_width; }
    uint32_t get_sqrt_spp() const { return desc_.sqrt_spp; }
    uint32_t get_max_depth() const { return desc_.max_depth; }
    const pt_camera_desc &desc() const { return desc_; }

    // Camera::render (src/camera.rs:162-236): renders every stratum on HIP device `device`
    // and leaves the tone-mapped image in `buffer`.  `progress` (optional) is called with
    // (passes done, passes total) between batches, where the reference refreshes its window.
    // Throws std::runtime_error carrying pt_last_error() on failure.
    void render(const PrimitiveBuffer &objects, const LightBuffer &lights, RgbImage &buffer, int device = 0, uint64_t seed = 0,
                void (*progress)(uint32_t, uint32_t, void *) = nullptr, void *user = nullptr) const;

private:
    pt_camera_desc desc_{};
};

}  // namespace pt
