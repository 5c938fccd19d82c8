// Host scene layer: see pthost.hpp.  Compiled with -ffp-contract=off; every f32
// expression below is meant to round exactly as the reference's does.
#include "pthost.hpp"

#include <algorithm>
#include <atomic>
#include <deque>
#include <list>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

#include "../../include/pthost.h"

namespace pt {

// ------------------------------------------------------------------ vectors, matrices
Vector3 operator+(Vector3 a, Vector3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
Vector3 operator-(Vector3 a, Vector3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Vector3 operator-(Vector3 a) { return {-a.x, -a.y, -a.z}; }
Vector3 operator*(float s, Vector3 a) { return {s * a.x, s * a.y, s * a.z}; }
Vector3 operator*(Vector3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
Vector3 operator/(Vector3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
float dot(Vector3 a, Vector3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
Vector3 cross(Vector3 a, Vector3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float norm(Vector3 a) { return std::sqrt(dot(a, a)); }
Vector3 normalize(Vector3 a) {
    const float n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}
static const float kPi = 3.14159265358979323846f;
static float to_radians(float deg) { return deg * (kPi / 180.0f); }
static float fmin_(float a, float b) { return std::fmin(a, b); }
static float fmax_(float a, float b) { return std::fmax(a, b); }

Matrix4 Matrix4::identity() {
    Matrix4 r{};
    for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f;
    return r;
}
// Product accumulated per output column over k = 0..3 (nalgebra's gemv order).
Matrix4 operator*(const Matrix4 &a, const Matrix4 &b) {
    Matrix4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float acc = a.m[i][0] * b.m[0][j];
            acc = a.m[i][1] * b.m[1][j] + acc;
            acc = a.m[i][2] * b.m[2][j] + acc;
            acc = a.m[i][3] * b.m[3][j] + acc;
            r.m[i][j] = acc;
        }
    return r;
}
// Rotation3::from_axis_angle(Unit::new_and_get(axisangle)).to_homogeneous()
Matrix4 Matrix4::new_rotation(Vector3 axisangle) {
    Matrix4 r = identity();
    const float angle = norm(axisangle);
    if (angle == 0.0f) return r;
    const float ux = axisangle.x / angle, uy = axisangle.y / angle, uz = axisangle.z / angle;
    const float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    const float sin = std::sin(angle), cos = std::cos(angle);
    const float one_m_cos = 1.0f - cos;
    r.m[0][0] = sqx + (1.0f - sqx) * cos;
    r.m[0][1] = ux * uy * one_m_cos - uz * sin;
    r.m[0][2] = ux * uz * one_m_cos + uy * sin;
    r.m[1][0] = ux * uy * one_m_cos + uz * sin;
    r.m[1][1] = sqy + (1.0f - sqy) * cos;
    r.m[1][2] = uy * uz * one_m_cos - ux * sin;
    r.m[2][0] = ux * uz * one_m_cos - uy * sin;
    r.m[2][1] = uy * uz * one_m_cos + ux * sin;
    r.m[2][2] = sqz + (1.0f - sqz) * cos;
    return r;
}
Matrix4 Matrix4::new_translation(Vector3 t) {
    Matrix4 r = identity();
    r.m[0][3] = t.x;
    r.m[1][3] = t.y;
    r.m[2][3] = t.z;
    return r;
}
Matrix4 Matrix4::new_nonuniform_scaling(Vector3 s) {
    Matrix4 r = identity();
    r.m[0][0] = s.x;
    r.m[1][1] = s.y;
    r.m[2][2] = s.z;
    return r;
}
Matrix4 Matrix4::new_scaling(float s) { return new_nonuniform_scaling(Vector3(s, s, s)); }
Vector3 Matrix4::transform_point(Point3 p) const {
    const float n = ((m[3][0] * p.x + m[3][1] * p.y) + m[3][2] * p.z) + m[3][3];
    Vector3 r;
    r.x = ((m[0][0] * p.x + m[0][1] * p.y) + m[0][2] * p.z) + m[0][3];
    r.y = ((m[1][0] * p.x + m[1][1] * p.y) + m[1][2] * p.z) + m[1][3];
    r.z = ((m[2][0] * p.x + m[2][1] * p.y) + m[2][2] * p.z) + m[2][3];
    if (n != 0.0f) return r / n;
    return r;
}

// ------------------------------------------------------------------ BBox
BBox BBox::from_corners(Point3 a, Point3 b) {
    BBox r;
    r.min = Point3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z));
    r.max = Point3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z));
    const float EPS = 0.0001f, padding = 0.0001f;  // BBox::pad, src/bvh.rs:51-65
    if (std::fabs(r.max.x - r.min.x) < EPS) { r.min.x -= padding; r.max.x += padding; }
    if (std::fabs(r.max.y - r.min.y) < EPS) { r.min.y -= padding; r.max.y += padding; }
    if (std::fabs(r.max.z - r.min.z) < EPS) { r.min.z -= padding; r.max.z += padding; }
    return r;
}
BBox BBox::empty() {
    const float inf = std::numeric_limits<float>::infinity();
    return BBox{Point3(inf, inf, inf), Point3(-inf, -inf, -inf)};
}
BBox BBox::point(Point3 p) { return BBox{p, p}; }
Point3 BBox::centroid() const { return (min + max) * 0.5f; }
BBox BBox::merge(const BBox &o) const {
    return BBox{Point3(fmin_(min.x, o.min.x), fmin_(min.y, o.min.y), fmin_(min.z, o.min.z)),
                Point3(fmax_(max.x, o.max.x), fmax_(max.y, o.max.y), fmax_(max.z, o.max.z))};
}
BBox BBox::enclose(Point3 p) const {
    return BBox{Point3(fmin_(min.x, p.x), fmin_(min.y, p.y), fmin_(min.z, p.z)), Point3(fmax_(max.x, p.x), fmax_(max.y, p.y), fmax_(max.z, p.z))};
}
float BBox::get_surface_area() const {
    const Vector3 e = max - min;
    return std::fabs((e.x * e.y + e.x * e.z) + e.y * e.z) * 2.0f;  // extent.xxy().dot(extent.yzz())
}
int BBox::longest_axis() const {
    const Vector3 e = max - min;
    if (e.x > e.y && e.x > e.z) return 0;
    if (e.y > e.z) return 1;
    return 2;
}
Vector3 BBox::offset(Point3 p) const {
    Vector3 o = p - min;
    if (max.x > min.x) o.x /= max.x - min.x;
    if (max.y > min.y) o.y /= max.y - min.y;
    if (max.z > min.z) o.z /= max.z - min.z;
    return o;
}
BBox BBox::transform(const Matrix4 &t) const {
    const Point3 a1 = t.transform_point(min), a2 = t.transform_point(max);
    const Point3 b1 = t.transform_point(Point3(max.x, min.y, min.z)), b2 = t.transform_point(Point3(min.x, max.y, max.z));
    const Point3 c1 = t.transform_point(Point3(max.x, max.y, min.z)), c2 = t.transform_point(Point3(min.x, min.y, max.z));
    const Point3 d1 = t.transform_point(Point3(min.x, max.y, min.z)), d2 = t.transform_point(Point3(max.x, min.y, max.z));
    return from_corners(a1, a2).merge(from_corners(b1, b2)).merge(from_corners(c1, c2)).merge(from_corners(d1, d2));
}

// ------------------------------------------------------------------ textures, materials
bool RgbImage::save_ppm(const std::string &path) const {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    const bool ok = std::fwrite(data.data(), 1, data.size(), f) == data.size();
    std::fclose(f);
    return ok;
}
namespace Solid {
TextureRef make(Color albedo) {
    auto t = std::make_shared<Texture>();
    t->type = PTS_TEX_SOLID;
    t->albedo = albedo;
    return t;
}
}  // namespace Solid
namespace Checkered {
TextureRef make(TextureRef odd, TextureRef even, float inv_scale) {
    auto t = std::make_shared<Texture>();
    t->type = PTS_TEX_CHECKERED;
    t->odd = std::move(odd);
    t->even = std::move(even);
    t->inv_scale = inv_scale;
    return t;
}
TextureRef new_solid(Color odd, Color even, float inv_scale) { return make(Solid::make(odd), Solid::make(even), inv_scale); }
}  // namespace Checkered
namespace Image {
TextureRef make(std::shared_ptr<const RgbImage> image) {
    auto t = std::make_shared<Texture>();
    t->type = PTS_TEX_IMAGE;
    t->image = std::move(image);
    return t;
}
}  // namespace Image
namespace Noise {
TextureRef make(Color color, float scale, uint32_t seed) {
    auto t = std::make_shared<Texture>();
    t->type = PTS_TEX_NOISE;
    t->albedo = color;
    t->inv_scale = scale;
    t->seed = seed;
    return t;
}
}  // namespace Noise

static std::shared_ptr<Material> new_mat(uint32_t type, TextureRef tex) {
    auto m = std::make_shared<Material>();
    m->type = type;
    m->texture = std::move(tex);
    return m;
}
// DisneyMetal::new / DisneyGlass::new, src/materials/disney.rs:537-553, :721-737
static void set_alpha(Material &m, float roughness, float anisotropic) {
    const float A_MIN = 0.0001f;
    const float aspect = std::sqrt(1.0f - 0.9f * anisotropic);
    const float inv_aspect = 1.0f / aspect;
    const float a_x = fmax_(roughness * roughness * inv_aspect, A_MIN);
    const float a_z = fmax_(roughness * roughness * aspect, A_MIN);
    m.derived[0] = a_x;
    m.derived[1] = a_z;
    m.derived[2] = 1.0f / a_x;
    m.derived[3] = 1.0f / a_z;
}
// DisneyClearcoat::new, :403-413
static void set_clearcoat(Material &m, float gloss) {
    const float a = (1.0f - gloss) * 0.1f + (gloss * 0.001f);
    const float a2 = a * a;
    m.derived[4] = a2;
    m.derived[5] = 1.0f / std::log2(a2);
}
namespace Lambertian {
MaterialRef make(Color albedo) { return new_mat(PTS_MAT_LAMBERTIAN, Solid::make(albedo)); }
MaterialRef new_texture(TextureRef texture) { return new_mat(PTS_MAT_LAMBERTIAN, texture ? texture : Solid::make(Color(0.5f, 0.5f, 0.5f))); }
}  // namespace Lambertian
namespace Metal {
MaterialRef make(Color albedo, float fuzz) {
    auto m = new_mat(PTS_MAT_METAL, nullptr);
    m->p[0] = albedo.x;
    m->p[1] = albedo.y;
    m->p[2] = albedo.z;
    m->p[3] = fuzz < 0.0f ? 0.0f : (fuzz > 1.0f ? 1.0f : fuzz);  // fuzz.clamp(0.0, 1.0)
    return m;
}
}  // namespace Metal
namespace Dielectric {
MaterialRef make(float ref_idx) {
    auto m = new_mat(PTS_MAT_DIELECTRIC, nullptr);
    m->p[0] = ref_idx;
    return m;
}
}  // namespace Dielectric
namespace Diffuse {
MaterialRef make(Color albedo) { return new_mat(PTS_MAT_DIFFUSE_LIGHT, Solid::make(albedo)); }
MaterialRef gray(float v) { return make(Color(v, v, v)); }
MaterialRef new_texture(TextureRef texture) { return new_mat(PTS_MAT_DIFFUSE_LIGHT, texture ? texture : Solid::make(Color(0.5f, 0.5f, 0.5f))); }
}  // namespace Diffuse
namespace Disney {
MaterialRef make(TextureRef base_color, float subsurface, float metallic, float specular_transmission, float specular, float specular_tint,
                 float roughness, float anisotropic, float sheen, float sheen_tint, float clearcoat, float clearcoat_gloss, float ior) {
    auto m = new_mat(PTS_MAT_DISNEY, std::move(base_color));
    const float v[12] = {subsurface, metallic, fmax_(specular_transmission, 0.0001f), specular, specular_tint, roughness,
                         anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior};
    std::memcpy(m->p, v, sizeof v);
    set_alpha(*m, roughness, anisotropic);
    set_clearcoat(*m, clearcoat_gloss);
    return m;
}
}  // namespace Disney
namespace DisneyDiffuse {
MaterialRef new_from_texture(TextureRef t, float roughness, float subsurface) {
    auto m = new_mat(PTS_MAT_DISNEY_DIFFUSE, std::move(t));
    m->p[0] = roughness;
    m->p[1] = subsurface;
    return m;
}
MaterialRef make(Color c, float roughness, float subsurface) { return new_from_texture(Solid::make(c), roughness, subsurface); }
}  // namespace DisneyDiffuse
namespace DisneyMetal {
MaterialRef new_from_texture(TextureRef t, float roughness, float anisotropic) {
    auto m = new_mat(PTS_MAT_DISNEY_METAL, std::move(t));
    m->p[0] = roughness;
    m->p[1] = anisotropic;
    set_alpha(*m, roughness, anisotropic);
    return m;
}
MaterialRef make(Color c, float roughness, float anisotropic) { return new_from_texture(Solid::make(c), roughness, anisotropic); }
}  // namespace DisneyMetal
namespace DisneyGlass {
MaterialRef new_from_texture(TextureRef t, float roughness, float anisotropic, float eta) {
    auto m = new_mat(PTS_MAT_DISNEY_GLASS, std::move(t));
    m->p[0] = roughness;
    m->p[1] = anisotropic;
    m->p[2] = eta;
    set_alpha(*m, roughness, anisotropic);
    return m;
}
MaterialRef make(Color c, float roughness, float anisotropic, float eta) { return new_from_texture(Solid::make(c), roughness, anisotropic, eta); }
}  // namespace DisneyGlass
namespace DisneyClearcoat {
MaterialRef make(float clearcoat, float clearcoat_gloss) {
    auto m = new_mat(PTS_MAT_DISNEY_CLEARCOAT, nullptr);
    m->p[0] = clearcoat;
    m->p[1] = clearcoat_gloss;
    set_clearcoat(*m, clearcoat_gloss);
    return m;
}
}  // namespace DisneyClearcoat
namespace DisneySheen {
MaterialRef new_from_texture(TextureRef t, float sheen, float sheen_tint) {
    auto m = new_mat(PTS_MAT_DISNEY_SHEEN, std::move(t));
    m->p[0] = sheen;
    m->p[1] = sheen_tint;
    return m;
}
MaterialRef make(Color c, float sheen, float sheen_tint) { return new_from_texture(Solid::make(c), sheen, sheen_tint); }
}  // namespace DisneySheen

MaterialRef MaterialRegistry::get(const std::string &name) const {
    auto it = materials_.find(name);
    if (it == materials_.end()) {
        std::fprintf(stderr, "[pthost] Material not found: %s\n", name.c_str());
        return nullptr;
    }
    return it->second;
}

// ------------------------------------------------------------------ primitives
Triangle::Triangle(const std::array<Point3, 3> &v, const std::array<Vector3, 3> *normals, const std::array<std::array<float, 2>, 3> *uv,
                   MaterialRef mat)
    : material(std::move(mat)) {
    std::array<Vector3, 3> n;
    if (normals) n = *normals;
    else {
        const Vector3 g = normalize(cross(v[1] - v[0], v[2] - v[0]));  // get_normal
        n = {g, g, g};
    }
    for (int c = 0; c < 3; c++) {
        data[0][c] = v[c].x; data[1][c] = v[c].y; data[2][c] = v[c].z;
        data[3][c] = n[c].x; data[4][c] = n[c].y; data[5][c] = n[c].z;
        data[6][c] = uv ? (*uv)[c][0] : 0.0f;
        data[7][c] = uv ? (*uv)[c][1] : 0.0f;
    }
}
BBox Triangle::bbox() const {
    const Point3 a = position(0), b = position(1), c = position(2);
    return BBox::from_corners(Point3(fmin_(a.x, fmin_(b.x, c.x)), fmin_(a.y, fmin_(b.y, c.y)), fmin_(a.z, fmin_(b.z, c.z))),
                              Point3(fmax_(a.x, fmax_(b.x, c.x)), fmax_(a.y, fmax_(b.y, c.y)), fmax_(a.z, fmax_(b.z, c.z))));
}
Sphere::Sphere(Point3 c, float r, MaterialRef m) : center(c), radius(r), mat(std::move(m)) {
    const Vector3 rvec(r, r, r);
    bbox_ = BBox::from_corners(c - rvec, c + rvec);
}
Quad::Quad(Point3 o, Vector3 u_, Vector3 v_, MaterialRef m) : origin(o), u(u_), v(v_), mat(std::move(m)) {
    const BBox diag1 = BBox::from_corners(o, o + u_ + v_);
    const BBox diag2 = BBox::from_corners(o + u_, o + v_);
    bbox_ = diag1.merge(diag2);
    const Vector3 n = cross(u_, v_);
    normal = normalize(n);
    d = dot(normal, o);
    w = n / dot(n, n);
    area = norm(n);
}
PrimitiveBuffer Quad::new_box(Point3 a, Point3 b, MaterialRef mat) {
    PrimitiveBuffer sides;
    const Point3 mn(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)), mx(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z));
    const Vector3 dx(mx.x - mn.x, 0.0f, 0.0f), dy(0.0f, mx.y - mn.y, 0.0f), dz(0.0f, 0.0f, mx.z - mn.z);
    sides.add_quad(Quad(Point3(mn.x, mn.y, mx.z), dx, dy, mat));
    sides.add_quad(Quad(Point3(mx.x, mn.y, mx.z), -dz, dy, mat));
    sides.add_quad(Quad(Point3(mx.x, mn.y, mn.z), -dx, dy, mat));
    sides.add_quad(Quad(Point3(mn.x, mn.y, mn.z), dz, dy, mat));
    sides.add_quad(Quad(Point3(mn.x, mx.y, mx.z), dx, -dz, mat));
    sides.add_quad(Quad(Point3(mn.x, mn.y, mn.z), dx, dz, mat));
    return sides;
}

Instance Instance::from_obj(PrimitiveBuffer obj) {
    Instance i;
    i.obj = std::make_shared<const PrimitiveBuffer>(std::move(obj));
    i.transform = Matrix4::identity();
    i.inverse = Matrix4::identity();
    return i;
}
// each operation left-multiplies `transform` and right-multiplies `inverse` (src/objects.rs:306-380)
void Instance::rotate(float angle, Vector3 axis) {
    transform = Matrix4::new_rotation(angle * axis) * transform;
    inverse = inverse * Matrix4::new_rotation(-angle * axis);
}
void Instance::rotate_x(float a) { transform = Matrix4::new_rotation(Vector3(a, 0, 0)) * transform; inverse = inverse * Matrix4::new_rotation(Vector3(-a, 0, 0)); }
void Instance::rotate_y(float a) { transform = Matrix4::new_rotation(Vector3(0, a, 0)) * transform; inverse = inverse * Matrix4::new_rotation(Vector3(0, -a, 0)); }
void Instance::rotate_z(float a) { transform = Matrix4::new_rotation(Vector3(0, 0, a)) * transform; inverse = inverse * Matrix4::new_rotation(Vector3(0, 0, -a)); }
void Instance::translate(Vector3 s) { transform = Matrix4::new_translation(s) * transform; inverse = inverse * Matrix4::new_translation(-s); }
void Instance::translate_x(float x) { transform = Matrix4::new_translation(Vector3(x, 0, 0)) * transform; inverse = inverse * Matrix4::new_translation(Vector3(-x, 0, 0)); }
void Instance::translate_y(float y) { transform = Matrix4::new_translation(Vector3(0, y, 0)) * transform; inverse = inverse * Matrix4::new_translation(Vector3(0, -y, 0)); }
void Instance::translate_z(float z) { transform = Matrix4::new_translation(Vector3(0, 0, z)) * transform; inverse = inverse * Matrix4::new_translation(Vector3(0, 0, -z)); }
void Instance::scale(Vector3 s) {
    transform = Matrix4::new_nonuniform_scaling(s) * transform;
    inverse = inverse * Matrix4::new_nonuniform_scaling(Vector3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z));
}
void Instance::scale_x(float x) { transform = Matrix4::new_nonuniform_scaling(Vector3(x, 1, 1)) * transform; inverse = inverse * Matrix4::new_nonuniform_scaling(Vector3(1.0f / x, 1, 1)); }
void Instance::scale_y(float y) { transform = Matrix4::new_nonuniform_scaling(Vector3(1, y, 1)) * transform; inverse = inverse * Matrix4::new_nonuniform_scaling(Vector3(1, 1.0f / y, 1)); }
void Instance::scale_z(float z) { transform = Matrix4::new_nonuniform_scaling(Vector3(1, 1, z)) * transform; inverse = inverse * Matrix4::new_nonuniform_scaling(Vector3(1, 1, 1.0f / z)); }
void Instance::scale_uniform(float s) { transform = Matrix4::new_scaling(s) * transform; inverse = inverse * Matrix4::new_scaling(1.0f / s); }
BBox Instance::bbox() const {  // src/objects.rs:407-409, level by level for an instance that was nested
    BBox b = obj->bbox;
    for (auto it = inner_levels.rbegin(); it != inner_levels.rend(); ++it) b = b.transform(it->first);
    return b.transform(transform);
}

void PrimitiveBuffer::add_triangle(Triangle t) { bbox = bbox.merge(t.bbox()); triangles.push(std::move(t)); }
void PrimitiveBuffer::add_sphere(Sphere s) { bbox = bbox.merge(s.bbox()); spheres.push(std::move(s)); }
void PrimitiveBuffer::add_quad(Quad q) { bbox = bbox.merge(q.bbox()); quads.push(std::move(q)); }
void PrimitiveBuffer::add_instance(Instance i) { bbox = bbox.merge(i.bbox()); instances.push(std::move(i)); }
void PrimitiveBuffer::build_bvh(size_t skip_tris_from) {
    const char *msg = "BVHBuilder::build failed: tree deeper than 63 levels";
    const bool skip = skip_tris_from != 0 && triangles.len() >= skip_tris_from;
    if (triangles.len() > 0 && !skip && !BVHBuilder::build(triangles)) throw std::runtime_error(msg);
    if (spheres.len() > 0 && !BVHBuilder::build(spheres)) throw std::runtime_error(msg);
    if (quads.len() > 0 && !BVHBuilder::build(quads)) throw std::runtime_error(msg);
    if (instances.len() > 0 && !BVHBuilder::build(instances)) throw std::runtime_error(msg);
}

// ------------------------------------------------------------------ BVH builder
namespace {
constexpr int kBuckets = 12;
constexpr size_t kMaxLeaf = 3;

// The tree is found first (large subtrees on their own threads: they work on disjoint parts of `order`)
// and written out in pre-order afterwards, the two steps of BVHBuilder::build_bvh and flatten_bvh
// (src/bvh.rs:313-516).  What each node decides depends on its own range only, so the result does not
// depend on the number of threads.
struct TreeNode {
    BBox bbox;
    uint32_t start = 0, size = 0;  // leaf: its range of `order`
    int axis = -1;                 // interior: split axis, children below
    TreeNode *left = nullptr, *right = nullptr;
};
// Nodes of one thread's subtrees (addresses stay valid while the deque grows).
using Arena = std::deque<TreeNode>;

struct BuildCtx {
    BuildCtx(const std::vector<BBox> &b, std::vector<uint32_t> &o) : boxes(b), order(o) {}
    const std::vector<BBox> &boxes;
    std::vector<Point3> centroids;  // per original primitive
    std::vector<uint32_t> &order;
    std::atomic<int> spare_threads{0};
    std::atomic<bool> failed{false};
    std::mutex arenas_lock;
    std::list<Arena> arenas;
    Arena &new_arena() {
        std::lock_guard<std::mutex> g(arenas_lock);
        arenas.emplace_back();
        return arenas.back();
    }
};

// `(12.0 * offset) as usize` then min(11): saturating conversion, NaN -> 0
int bucket_index(const BBox &cb, Point3 c, int axis) {
    const float f = (float)kBuckets * cb.offset(c)[axis];
    long long b = 0;
    if (f > 0.0f) b = f >= 9.0e18f ? (long long)9e18 : (long long)f;
    return (int)std::min<long long>(b, kBuckets - 1);
}

// The subtree over order[start, end); null (and c.failed) past depth 63 or on an empty side.
TreeNode *grow(BuildCtx &c, Arena &arena, size_t start, size_t end, uint32_t depth) {
    if (depth > 63 || c.failed.load(std::memory_order_relaxed)) {
        c.failed = true;
        return nullptr;
    }
    arena.emplace_back();
    TreeNode *node = &arena.back();
    BBox bbox = c.boxes[c.order[start]];
    for (size_t i = start; i < end; i++) bbox = bbox.merge(c.boxes[c.order[i]]);
    node->bbox = bbox;
    const size_t size = end - start;
    auto leaf = [&]() {
        node->start = (uint32_t)start;
        node->size = (uint32_t)size;
        return node;
    };
    if (size == 1) return leaf();
    const int axis = bbox.longest_axis();
    BBox cb = BBox::point(c.centroids[c.order[start]]);
    for (size_t i = start; i < end; i++) cb = cb.enclose(c.centroids[c.order[i]]);
    if (cb.max[axis] - cb.min[axis] <= 0.0001f) return leaf();

    size_t mid;
    if (size <= 2) {
        // stable sort of two elements by centroid on the axis
        if (c.centroids[c.order[start + 1]][axis] < c.centroids[c.order[start]][axis]) std::swap(c.order[start], c.order[start + 1]);
        mid = start + size / 2;
    } else {
        uint32_t count[kBuckets] = {0};
        bool used[kBuckets] = {false};
        BBox bb[kBuckets];
        for (size_t i = start; i < end; i++) {
            const uint32_t p = c.order[i];
            const int b = bucket_index(cb, c.centroids[p], axis);
            count[b]++;
            bb[b] = used[b] ? bb[b].merge(c.boxes[p]) : c.boxes[p];
            used[b] = true;
        }
        float cost[kBuckets - 1];
        {
            bool have = false;
            BBox acc{};
            uint32_t n = 0;
            for (int i = 0; i < kBuckets - 1; i++) {
                if (used[i]) { acc = have ? acc.merge(bb[i]) : bb[i]; have = true; }
                n += count[i];
                cost[i] = 0.0f + (float)n * (have ? acc.get_surface_area() : 0.0f);
            }
        }
        {
            bool have = false;
            BBox acc{};
            uint32_t n = 0;
            for (int i = kBuckets - 1; i >= 1; i--) {
                if (used[i]) { acc = have ? acc.merge(bb[i]) : bb[i]; have = true; }
                n += count[i];
                cost[i - 1] += (float)n * (have ? acc.get_surface_area() : 0.0f);
            }
        }
        float min_cost = std::numeric_limits<float>::infinity();
        int split = kBuckets / 2;
        for (int i = 0; i < kBuckets - 1; i++)
            if (cost[i] < min_cost) { min_cost = cost[i]; split = i; }
        min_cost = 0.5f + (((float)size * min_cost) / bbox.get_surface_area());
        const float leaf_cost = (float)size;
        if (!(size > kMaxLeaf || min_cost < leaf_cost)) return leaf();
        // partition_mut (src/bvh.rs:11-31), swap for swap
        auto pred = [&](uint32_t p) { return bucket_index(cb, c.centroids[p], axis) <= split; };
        size_t i = start, j = end - 1;
        while (i < j) {
            if (!pred(c.order[j])) { j -= 1; continue; }
            if (pred(c.order[i])) { i += 1; continue; }
            std::swap(c.order[i], c.order[j]);
        }
        mid = i + 1;
    }
    node->axis = axis;
    if (mid <= start || mid >= end) {
        // The reference would recurse into an empty range here and hit its `start < end` assertion.
        c.failed = true;
        return nullptr;
    }
    constexpr size_t kOwnThread = 1u << 14;  // smaller subtrees are not worth a thread
    if (mid - start >= kOwnThread && end - mid >= kOwnThread && c.spare_threads.fetch_sub(1) > 0) {
        // Nothing may leave this block by unwinding: a joinable std::thread that is destroyed, or an exception
        // that escapes a thread's function, ends the process.  A failure in either half marks the build failed.
        TreeNode *left = nullptr;
        std::thread t;
        try {
            t = std::thread([&] {
                try {
                    left = grow(c, c.new_arena(), start, mid, depth + 1);
                } catch (...) {
                    c.failed = true;
                    left = nullptr;
                }
            });
        } catch (...) {  // no thread to be had: this one does both halves
        }
        const bool threaded = t.joinable();
        try {
            node->right = grow(c, arena, mid, end, depth + 1);
            if (!threaded) left = grow(c, arena, start, mid, depth + 1);
        } catch (...) {
            c.failed = true;
            node->right = nullptr;
        }
        if (threaded) t.join();
        c.spare_threads.fetch_add(1);
        node->left = left;
    } else {
        if (mid - start >= kOwnThread && end - mid >= kOwnThread) c.spare_threads.fetch_add(1);  // undo the probe
        node->left = grow(c, arena, start, mid, depth + 1);
        node->right = grow(c, arena, mid, end, depth + 1);
    }
    if (!node->left || !node->right) return nullptr;
    return node;
}

pt_node make_node(const BBox &b, uint32_t a, uint32_t bb) {
    pt_node n;
    n.bmin[0] = b.min.x; n.bmin[1] = b.min.y; n.bmin[2] = b.min.z;
    n.bmax[0] = b.max.x; n.bmax[1] = b.max.y; n.bmax[2] = b.max.z;
    n.a = a;
    n.b = bb;
    return n;
}

// flatten_bvh: pre-order, the left child follows its parent, `a` of an interior node is its right child.
void flatten(const TreeNode &root, std::vector<pt_node> &nodes) {
    struct Item {
        const TreeNode *node;
        int64_t parent;  // interior node whose right child this is, or -1
    };
    std::vector<Item> stack{{&root, -1}};
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        if (it.parent >= 0) nodes[(size_t)it.parent].a = (uint32_t)nodes.size();
        const TreeNode &n = *it.node;
        if (n.axis < 0) {
            nodes.push_back(make_node(n.bbox, n.start, PT_NODE_LEAF | (uint32_t)(uint16_t)n.size));
        } else {
            const int64_t self = (int64_t)nodes.size();
            nodes.push_back(make_node(n.bbox, 0u, (uint32_t)n.axis));
            stack.push_back({n.right, self});
            stack.push_back({n.left, -1});
        }
    }
}
}  // namespace

bool BVHBuilder::build(const std::vector<BBox> &boxes, std::vector<uint32_t> &order, std::vector<pt_node> &nodes) {
    nodes.clear();
    if (boxes.empty()) return true;
    BuildCtx c(boxes, order);
    c.centroids.reserve(boxes.size());
    for (const BBox &b : boxes) c.centroids.push_back(b.centroid());
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = std::getenv("PTH_BUILD_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));  // 1 = no extra threads
    c.spare_threads = (int)std::min(15u, hw > 1 ? hw - 1 : 0u);
    TreeNode *root = grow(c, c.new_arena(), 0, boxes.size(), 0);
    if (!root || c.failed) return false;
    nodes.reserve(2 * boxes.size());
    flatten(*root, nodes);
    return true;
}

template <class T>
bool BVHBuilder::build(InnerPrimitiveBuffer<T> &objects) {
    const size_t n = objects.buffer.size();
    std::vector<BBox> boxes;
    boxes.reserve(n);
    for (const T &p : objects.buffer) boxes.push_back(p.bbox());
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    std::vector<pt_node> nodes;
    if (!build(boxes, order, nodes)) return false;
    std::vector<T> sorted;
    sorted.reserve(n);
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++) {
        sorted.push_back(std::move(objects.buffer[order[i]]));
        perm[i] = objects.perm[order[i]];
    }
    objects.buffer = std::move(sorted);
    objects.perm = std::move(perm);
    objects.bvh = std::move(nodes);
    return true;
}
template bool BVHBuilder::build(InnerPrimitiveBuffer<Triangle> &);
template bool BVHBuilder::build(InnerPrimitiveBuffer<Sphere> &);
template bool BVHBuilder::build(InnerPrimitiveBuffer<Quad> &);
template bool BVHBuilder::build(InnerPrimitiveBuffer<Instance> &);

// ------------------------------------------------------------------ flattening
int FlatScene::texture_index(const TextureRef &t) {
    if (!t) return -1;
    auto it = tex_ids_.find(t.get());
    if (it != tex_ids_.end()) return it->second;
    pts_texture d{};
    d.type = t->type;
    d.odd = d.even = d.image = -1;
    if (t->type == PTS_TEX_SOLID) {
        d.color[0] = t->albedo.x; d.color[1] = t->albedo.y; d.color[2] = t->albedo.z;
    } else if (t->type == PTS_TEX_NOISE) {
        d.color[0] = t->albedo.x; d.color[1] = t->albedo.y; d.color[2] = t->albedo.z;
        d.scale = t->inv_scale;
        d.odd = (int32_t)t->seed;
    } else if (t->type == PTS_TEX_CHECKERED) {
        d.odd = texture_index(t->odd);  // children first: their indices precede the parent's
        d.even = texture_index(t->even);
        d.scale = t->inv_scale;
    } else if (t->image) {
        auto ii = img_ids_.find(t->image.get());
        if (ii == img_ids_.end()) {
            pts_image im{t->image->width, t->image->height, texels_.size()};
            texels_.insert(texels_.end(), t->image->data.begin(), t->image->data.end());
            images_.push_back(im);
            ii = img_ids_.emplace(t->image.get(), (int)images_.size() - 1).first;
        }
        d.image = ii->second;
    }
    textures_.push_back(d);
    keep_t_.push_back(t);
    const int id = (int)textures_.size() - 1;
    tex_ids_[t.get()] = id;
    return id;
}
int FlatScene::material_index(const MaterialRef &m) {
    if (!m) return -1;
    auto it = mat_ids_.find(m.get());
    if (it != mat_ids_.end()) return it->second;
    pt_material_desc d{};
    d.type = m->type;
    d.texture = texture_index(m->texture);
    std::memcpy(d.p, m->p, sizeof d.p);
    std::memcpy(d.derived, m->derived, sizeof d.derived);
    materials_.push_back(d);
    keep_m_.push_back(m);
    const int id = (int)materials_.size() - 1;
    mat_ids_[m.get()] = id;
    return id;
}
template <class T>
pt_buffer_desc FlatScene::add_nodes(const InnerPrimitiveBuffer<T> &b, uint32_t prim_first) {
    pt_buffer_desc d{prim_first, (uint32_t)b.buffer.size(), (uint32_t)nodes_.size(), (uint32_t)b.bvh.size()};
    nodes_.insert(nodes_.end(), b.bvh.begin(), b.bvh.end());
    return d;
}
// instances: the instance buffer to describe for this set; null = none (the set of an instance: the device
// library searches two levels, and build() has lifted deeper instances to the top one).
uint32_t FlatScene::add_set(const PrimitiveBuffer &pb, const InnerPrimitiveBuffer<Instance> *instances) {
    const uint32_t index = (uint32_t)sets_.size();
    sets_.push_back(pt_primset_desc{});
    pt_primset_desc d{};
    d.tris = add_nodes(pb.triangles, (uint32_t)tri_mat_.size());
    for (const Triangle &t : pb.triangles.buffer) {
        tri_data_.insert(tri_data_.end(), &t.data[0][0], &t.data[0][0] + 24);
        tri_mat_.push_back(material_index(t.material));
    }
    d.spheres = add_nodes(pb.spheres, (uint32_t)sphere_mat_.size());
    for (const Sphere &s : pb.spheres.buffer) {
        const float v[4] = {s.center.x, s.center.y, s.center.z, s.radius};
        spheres_.insert(spheres_.end(), v, v + 4);
        sphere_mat_.push_back(material_index(s.mat));
    }
    d.quads = add_nodes(pb.quads, (uint32_t)quad_mat_.size());
    for (const Quad &q : pb.quads.buffer) {
        const float v[17] = {q.origin.x, q.origin.y, q.origin.z, q.u.x, q.u.y, q.u.z, q.v.x, q.v.y, q.v.z,
                             q.normal.x, q.normal.y, q.normal.z, q.w.x, q.w.y, q.w.z, q.d, q.area};
        quads_.insert(quads_.end(), v, v + 17);
        quad_mat_.push_back(material_index(q.mat));
    }
    if (!instances) {
        d.instances = pt_buffer_desc{(uint32_t)instances_.size(), 0u, (uint32_t)nodes_.size(), 0u};
        sets_[index] = d;
        return index;
    }
    d.instances = add_nodes(*instances, (uint32_t)instances_.size());
    const size_t inst_first = instances_.size();
    instances_.resize(inst_first + instances->buffer.size());
    sets_[index] = d;
    for (size_t k = 0; k < instances->buffer.size(); k++) {
        const Instance &in = instances->buffer[k];
        pt_instance_desc id{};
        std::memcpy(id.transform, in.transform.m, sizeof id.transform);
        std::memcpy(id.inverse, in.inverse.m, sizeof id.inverse);
        id.child_set = 0u;  // patched by build() once the child set has an index
        id.chain_first = (uint32_t)chain_.size();  // within chain_; build() moves the records behind every set's instances
        id.chain_count = (uint32_t)in.inner_levels.size();
        for (const auto &level : in.inner_levels) {
            pt_instance_desc step{};
            std::memcpy(step.transform, level.first.m, sizeof step.transform);
            std::memcpy(step.inverse, level.second.m, sizeof step.inverse);
            chain_.push_back(step);
        }
        instances_[inst_first + k] = id;
    }
    return index;
}
void FlatScene::add_medium(Point3 center, float radius, float density, const TextureRef &albedo) {
    if (media_.size() >= PTS_MAX_MEDIA) throw std::runtime_error("too many participating media");
    pts_medium m{};
    m.center[0] = center.x; m.center[1] = center.y; m.center[2] = center.z;
    m.radius = radius;
    m.density = density;
    m.texture = texture_index(albedo);
    media_.push_back(m);
}

void FlatScene::add_moving_sphere(Point3 center0, Point3 center1, float radius, const MaterialRef &material) {
    if (moving_.size() >= PTS_MAX_MOVING) throw std::runtime_error("too many moving spheres");
    pts_moving_sphere m{};
    m.center0[0] = center0.x; m.center0[1] = center0.y; m.center0[2] = center0.z;
    m.center1[0] = center1.x; m.center1[1] = center1.y; m.center1[2] = center1.z;
    m.radius = radius;
    m.material = material_index(material);
    moving_.push_back(m);
}

// An Instance holds a whole PrimitiveBuffer (src/objects.rs:281-286), which may hold instances again.  The device
// library searches two levels, so an instance inside an instance is listed at the top level, once for every way
// down to it, and keeps the levels of its nesting: Instance::hit of the outermost one takes the ray through its
// `inverse`, the next one through its own, and so on (src/objects.rs:385-390), and the way back goes through the
// `transform`s in the opposite order (:393-394).  The device applies them one at a time in that order, so distances,
// hit points and normals are the nested evaluation's to the bit.  What may differ: which of several hits at exactly
// the same distance wins, because the lifted instances are visited in the order of their own place in the top
// level's instance buffer (and its BVH, rebuilt over the longer list) rather than inside the outer instance.
static void lift_instances(const Instance &in, std::vector<std::pair<Matrix4, Matrix4>> &levels, InnerPrimitiveBuffer<Instance> &out) {
    if (levels.size() > 8) throw std::runtime_error("instance nesting too deep");
    levels.emplace_back(in.transform, in.inverse);
    const PrimitiveBuffer &child = *in.obj;
    if (child.triangles.len() + child.spheres.len() + child.quads.len() > 0 || child.instances.len() == 0) {
        Instance flat;
        flat.obj = in.obj;  // described without its instance buffer
        flat.transform = levels[0].first;
        flat.inverse = levels[0].second;
        flat.inner_levels.assign(levels.begin() + 1, levels.end());
        out.push(std::move(flat));
    }
    for (const Instance &inner : child.instances.buffer) lift_instances(inner, levels, out);
    levels.pop_back();
}

void FlatScene::build(const PrimitiveBuffer &objects, const LightBuffer &lights) {
    bool nested = false;
    for (const Instance &in : objects.instances.buffer) nested = nested || in.obj->instances.len() > 0;
    InnerPrimitiveBuffer<Instance> lifted;
    if (nested) {
        std::vector<std::pair<Matrix4, Matrix4>> levels;
        for (const Instance &in : objects.instances.buffer) lift_instances(in, levels, lifted);
        if (!objects.instances.bvh.empty() && !BVHBuilder::build(lifted)) throw std::runtime_error("BVHBuilder::build failed: tree deeper than 63 levels");
    }
    const InnerPrimitiveBuffer<Instance> &top = nested ? lifted : objects.instances;
    add_set(objects, &top);
    add_set(lights.objects, &lights.objects.instances);
    // children of the world's instances, in post-build instance order
    const pt_buffer_desc wi = sets_[0].instances;
    for (uint32_t k = 0; k < wi.prim_count; k++) {
        const uint32_t child = add_set(*top.buffer[k].obj, nullptr);
        instances_[wi.prim_first + k].child_set = child;
    }
    // instances listed as lights keep a child set too, although the integrator never enters it
    const pt_buffer_desc li = sets_[1].instances;
    for (uint32_t k = 0; k < li.prim_count; k++) {
        const uint32_t child = add_set(*lights.objects.instances.buffer[k].obj, nullptr);
        instances_[li.prim_first + k].child_set = child;
    }
    // the levels of nested instances go behind every set's instances
    const uint32_t chain_base = (uint32_t)instances_.size();
    for (uint32_t k = 0; k < wi.prim_count; k++) {
        pt_instance_desc &id = instances_[wi.prim_first + k];
        for (uint32_t j = 0; j < id.chain_count; j++) chain_[id.chain_first + j].child_set = id.child_set;
        id.chain_first = id.chain_count ? id.chain_first + chain_base : 0u;
    }
    for (uint32_t k = 0; k < li.prim_count; k++) instances_[li.prim_first + k].chain_first = instances_[li.prim_first + k].chain_count = 0u;
    instances_.insert(instances_.end(), chain_.begin(), chain_.end());
    // moving spheres go behind the spheres of the sets, as they are at time 0
    for (const pts_moving_sphere &m : moving_) {
        spheres_.insert(spheres_.end(), {m.center0[0], m.center0[1], m.center0[2], m.radius});
        sphere_mat_.push_back(m.material);
        for (int k = 0; k < 3; k++) moving_delta_.push_back(m.center1[k] - m.center0[k]);
    }
    pt_scene_desc d{};
    d.n_tris = (uint32_t)tri_mat_.size(); d.tri_data = tri_data_.data(); d.tri_mat = tri_mat_.data();
    d.n_spheres = (uint32_t)sphere_mat_.size(); d.spheres = spheres_.data(); d.sphere_mat = sphere_mat_.data();
    d.n_quads = (uint32_t)quad_mat_.size(); d.quads = quads_.data(); d.quad_mat = quad_mat_.data();
    d.n_instances = (uint32_t)instances_.size(); d.instances = instances_.data();
    d.n_nodes = (uint32_t)nodes_.size(); d.nodes = nodes_.data();
    d.n_sets = (uint32_t)sets_.size(); d.sets = sets_.data();
    d.n_materials = (uint32_t)materials_.size(); d.materials = materials_.data();
    d.n_textures = (uint32_t)textures_.size(); d.textures = textures_.data();
    d.n_images = (uint32_t)images_.size(); d.images = images_.data();
    d.n_texel_bytes = texels_.size(); d.texels = texels_.data();
    d.n_media = (uint32_t)media_.size(); d.media = media_.data();
    d.n_moving = (uint32_t)moving_.size(); d.moving = moving_delta_.data();
    desc_ = d;
}

// ------------------------------------------------------------------ camera
static uint32_t f2u32(float f) {  // Rust `as u32`
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
Camera::Camera(float aspect_ratio, uint32_t image_width, float vfov, Point3 look_from, Point3 look_at, float focal_length, float defocus_angle,
               uint32_t spp, uint32_t max_depth, Color background) {
    const uint32_t image_height = std::max(1u, f2u32((float)image_width / aspect_ratio));
    const float theta = to_radians(vfov);
    const float h = std::tan(theta / 2.0f);
    const Vector3 vup(0.0f, 1.0f, 0.0f);
    const Point3 center = look_from;
    const float viewport_height = 2.0f * h * focal_length;
    const float viewport_width = viewport_height * ((float)image_width / (float)image_height);
    const Vector3 w = normalize(look_from - look_at);
    const Vector3 u = normalize(cross(vup, w));
    const Vector3 v = cross(w, u);
    const Vector3 viewport_u = viewport_width * u;
    const Vector3 viewport_v = viewport_height * -v;
    const Vector3 pixel_du = viewport_u / (float)image_width;
    const Vector3 pixel_dv = viewport_v / (float)image_height;
    const Point3 viewport_upper_left = center - viewport_u / 2.0f - viewport_v / 2.0f - focal_length * w;
    const Point3 pixel00 = viewport_upper_left + 0.5f * pixel_du + 0.5f * pixel_dv;
    const float defocus_radius = focal_length * std::tan(0.5f * to_radians(defocus_angle / 2.0f));
    const Vector3 disk_u = u * defocus_radius, disk_v = v * defocus_radius;
    const uint32_t sqrt_spp = f2u32(std::sqrt((float)spp));
    pt_camera_desc &d = desc_;
    d.image_width = image_width;
    d.image_height = image_height;
    auto put = [](float *o, Vector3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; };
    put(d.center, center);
    put(d.pixel00, pixel00);
    put(d.pixel_du, pixel_du);
    put(d.pixel_dv, pixel_dv);
    put(d.defocus_disk_u, disk_u);
    put(d.defocus_disk_v, disk_v);
    d.defocus_angle = defocus_angle;
    d.sqrt_spp = sqrt_spp;
    d.inv_sqrt_spp = 1.0f / (float)sqrt_spp;
    d.max_depth = max_depth;
    put(d.background, background);
}

void Camera::render(const PrimitiveBuffer &objects, const LightBuffer &lights, RgbImage &buffer, int device, uint64_t seed,
                    void (*progress)(uint32_t, uint32_t, void *), void *user) const {
    FlatScene flat;
    flat.build(objects, lights);
    pt_context *ctx = nullptr;
    if (pt_create(device, &ctx) != PT_OK) throw std::runtime_error(std::string("pt_create: ") + pt_last_error(nullptr));
    auto die = [&](const char *what) {
        std::string msg = std::string(what) + ": " + pt_last_error(ctx);
        pt_destroy(ctx);
        throw std::runtime_error(msg);
    };
    if (pt_upload_scene(ctx, &flat.desc()) != PT_OK) die("pt_upload_scene");
    if (pt_set_camera(ctx, &desc_) != PT_OK) die("pt_set_camera");
    buffer = RgbImage(desc_.image_width, desc_.image_height);
    const uint32_t total = desc_.sqrt_spp * desc_.sqrt_spp;
    // progressive display: hand the image back every `chunk` passes
    const uint32_t chunk = progress ? std::max(1u, total / 16u) : total;
    for (uint32_t s = 0; s < total; s += chunk) {
        const uint32_t e = std::min(total, s + chunk);
        if (pt_render(ctx, s, e, seed) != PT_OK) die("pt_render");
        if (progress) {
            if (pt_read_rgb8(ctx, buffer.data.data()) != PT_OK) die("pt_read_rgb8");
            progress(e, total, user);
        }
    }
    if (pt_read_rgb8(ctx, buffer.data.data()) != PT_OK) die("pt_read_rgb8");
    pt_destroy(ctx);
}

}  // namespace pt

// ====================================================================== C entry points
struct pth_scene {
    pt::PrimitiveBuffer objects;
    pt::LightBuffer lights;
    pt::FlatScene flat;
    std::unique_ptr<pt::Camera> camera;
};

namespace {
using namespace pt;

struct SrcLoader {
    const pts_scene &s;
    std::vector<TextureRef> tex;
    std::vector<MaterialRef> mat;
    std::vector<std::shared_ptr<const RgbImage>> img;
    size_t skip_tris_from = 0;  // PrimitiveBuffer::build_bvh

    explicit SrcLoader(const pts_scene &sc) : s(sc) {}

    void load_tables() {
        for (uint32_t i = 0; i < s.n_images; i++) {
            auto im = std::make_shared<RgbImage>(s.images[i].width, s.images[i].height);
            const size_t n = (size_t)im->width * im->height * 3;
            if (s.images[i].offset + n > s.n_texel_bytes) throw std::runtime_error("image texels out of range");
            std::memcpy(im->data.data(), s.texels + s.images[i].offset, n);
            img.push_back(im);
        }
        tex.resize(s.n_textures);
        for (uint32_t i = 0; i < s.n_textures; i++) {
            const pts_texture &t = s.textures[i];
            switch (t.type) {
                case PTS_TEX_SOLID: tex[i] = Solid::make(Color(t.color[0], t.color[1], t.color[2])); break;
                case PTS_TEX_CHECKERED:
                    if (t.odd < 0 || t.even < 0 || (uint32_t)t.odd >= i || (uint32_t)t.even >= i) throw std::runtime_error("checker children must precede it");
                    tex[i] = Checkered::make(tex[t.odd], tex[t.even], t.scale);
                    break;
                case PTS_TEX_NOISE: tex[i] = Noise::make(Color(t.color[0], t.color[1], t.color[2]), t.scale, (uint32_t)t.odd); break;
                case PTS_TEX_IMAGE:
                    if (t.image >= (int32_t)img.size()) throw std::runtime_error("image index out of range");
                    tex[i] = Image::make(t.image >= 0 ? img[t.image] : nullptr);
                    break;
                default: throw std::runtime_error("unknown texture type");
            }
        }
        auto T = [&](int32_t id) -> TextureRef {
            if (id < 0 || id >= (int32_t)tex.size()) throw std::runtime_error("texture index out of range");
            return tex[id];
        };
        for (uint32_t i = 0; i < s.n_materials; i++) {
            const pts_material &m = s.materials[i];
            const float *p = m.p;
            switch (m.type) {
                case PTS_MAT_LAMBERTIAN: mat.push_back(Lambertian::new_texture(T(m.texture))); break;
                case PTS_MAT_METAL: mat.push_back(Metal::make(Color(p[0], p[1], p[2]), p[3])); break;
                case PTS_MAT_DIELECTRIC: mat.push_back(Dielectric::make(p[0])); break;
                case PTS_MAT_DIFFUSE_LIGHT: mat.push_back(Diffuse::new_texture(T(m.texture))); break;
                case PTS_MAT_DISNEY: mat.push_back(Disney::make(T(m.texture), p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11])); break;
                case PTS_MAT_DISNEY_DIFFUSE: mat.push_back(DisneyDiffuse::new_from_texture(T(m.texture), p[0], p[1])); break;
                case PTS_MAT_DISNEY_METAL: mat.push_back(DisneyMetal::new_from_texture(T(m.texture), p[0], p[1])); break;
                case PTS_MAT_DISNEY_GLASS: mat.push_back(DisneyGlass::new_from_texture(T(m.texture), p[0], p[1], p[2])); break;
                case PTS_MAT_DISNEY_CLEARCOAT: mat.push_back(DisneyClearcoat::make(p[0], p[1])); break;
                case PTS_MAT_DISNEY_SHEEN: mat.push_back(DisneySheen::new_from_texture(T(m.texture), p[0], p[1])); break;
                default: throw std::runtime_error("unknown material type");
            }
        }
    }
    MaterialRef M(int32_t id) const {
        if (id < 0) return nullptr;
        if (id >= (int32_t)mat.size()) throw std::runtime_error("material index out of range");
        return mat[id];
    }
    void load_set(uint32_t index, PrimitiveBuffer &out, int depth) {
        if (index >= s.n_sets) throw std::runtime_error("set index out of range");
        if (depth > 8) throw std::runtime_error("instance nesting too deep");
        const pts_primset &ps = s.sets[index];
        if ((uint64_t)ps.tris.first + ps.tris.count > s.n_tris || (uint64_t)ps.spheres.first + ps.spheres.count > s.n_spheres ||
            (uint64_t)ps.quads.first + ps.quads.count > s.n_quads || (uint64_t)ps.instances.first + ps.instances.count > s.n_instances)
            throw std::runtime_error("set range outside the scene arrays");
        for (uint32_t i = ps.tris.first; i < ps.tris.first + ps.tris.count; i++) {
            std::array<Point3, 3> v;
            std::array<Vector3, 3> n;
            std::array<std::array<float, 2>, 3> uv;
            for (int c = 0; c < 3; c++) v[c] = Point3(s.tri_pos[9 * (size_t)i + 3 * c], s.tri_pos[9 * (size_t)i + 3 * c + 1], s.tri_pos[9 * (size_t)i + 3 * c + 2]);
            const uint8_t flags = s.tri_flags ? s.tri_flags[i] : (uint8_t)((s.tri_nrm ? PTS_TRI_HAS_NORMALS : 0) | (s.tri_uv ? PTS_TRI_HAS_UVS : 0));
            const bool hn = (flags & PTS_TRI_HAS_NORMALS) && s.tri_nrm, hu = (flags & PTS_TRI_HAS_UVS) && s.tri_uv;
            if (hn) for (int c = 0; c < 3; c++) n[c] = Vector3(s.tri_nrm[9 * (size_t)i + 3 * c], s.tri_nrm[9 * (size_t)i + 3 * c + 1], s.tri_nrm[9 * (size_t)i + 3 * c + 2]);
            if (hu) for (int c = 0; c < 3; c++) uv[c] = {s.tri_uv[6 * (size_t)i + 2 * c], s.tri_uv[6 * (size_t)i + 2 * c + 1]};
            out.add_triangle(Triangle(v, hn ? &n : nullptr, hu ? &uv : nullptr, M(s.tri_mat[i])));
        }
        for (uint32_t i = ps.spheres.first; i < ps.spheres.first + ps.spheres.count; i++)
            out.add_sphere(Sphere(Point3(s.spheres[4 * (size_t)i], s.spheres[4 * (size_t)i + 1], s.spheres[4 * (size_t)i + 2]), s.spheres[4 * (size_t)i + 3], M(s.sphere_mat[i])));
        for (uint32_t i = ps.quads.first; i < ps.quads.first + ps.quads.count; i++) {
            const float *q = s.quads + 9 * (size_t)i;
            out.add_quad(Quad(Point3(q[0], q[1], q[2]), Vector3(q[3], q[4], q[5]), Vector3(q[6], q[7], q[8]), M(s.quad_mat[i])));
        }
        for (uint32_t i = ps.instances.first; i < ps.instances.first + ps.instances.count; i++) {
            const pts_instance &pi = s.instances[i];
            PrimitiveBuffer child;
            load_set(pi.child_set, child, depth + 1);
            Instance inst = Instance::from_obj(std::move(child));
            if ((uint64_t)pi.op_first + pi.op_count > s.n_ops) throw std::runtime_error("transform op range out of bounds");
            for (uint32_t k = 0; k < pi.op_count; k++) {
                const pts_xform_op &op = s.ops[pi.op_first + k];
                const float *a = op.a;
                switch (op.op) {
                    case PTS_OP_ROTATE: inst.rotate(a[0], Vector3(a[1], a[2], a[3])); break;
                    case PTS_OP_ROTATE_X: inst.rotate_x(a[0]); break;
                    case PTS_OP_ROTATE_Y: inst.rotate_y(a[0]); break;
                    case PTS_OP_ROTATE_Z: inst.rotate_z(a[0]); break;
                    case PTS_OP_TRANSLATE: inst.translate(Vector3(a[0], a[1], a[2])); break;
                    case PTS_OP_TRANSLATE_X: inst.translate_x(a[0]); break;
                    case PTS_OP_TRANSLATE_Y: inst.translate_y(a[0]); break;
                    case PTS_OP_TRANSLATE_Z: inst.translate_z(a[0]); break;
                    case PTS_OP_SCALE: inst.scale(Vector3(a[0], a[1], a[2])); break;
                    case PTS_OP_SCALE_X: inst.scale_x(a[0]); break;
                    case PTS_OP_SCALE_Y: inst.scale_y(a[0]); break;
                    case PTS_OP_SCALE_Z: inst.scale_z(a[0]); break;
                    case PTS_OP_SCALE_UNIFORM: inst.scale_uniform(a[0]); break;
                    default: throw std::runtime_error("unknown transform op");
                }
            }
            out.add_instance(std::move(inst));
        }
        if (ps.build_bvh) out.build_bvh(skip_tris_from);
    }
};
}  // namespace

extern "C" {

pth_scene *pth_scene_build(const pts_scene *scene, const pts_camera *camera, char *err, size_t errlen) {
    return pth_scene_build_opts(scene, camera, 0, err, errlen);
}

pth_scene *pth_scene_build_opts(const pts_scene *scene, const pts_camera *camera, uint32_t skip_tri_bvh_from, char *err, size_t errlen) {
    auto put_err = [&](const char *msg) {
        if (err && errlen) {
            std::strncpy(err, msg, errlen - 1);
            err[errlen - 1] = 0;
        }
    };
    if (!scene || !camera) {
        put_err("NULL argument");
        return nullptr;
    }
    try {
        if (scene->n_sets < 2) throw std::runtime_error("need at least the object set and the light set");
        auto out = std::make_unique<pth_scene>();
        SrcLoader L(*scene);
        L.skip_tris_from = skip_tri_bvh_from;
        L.load_tables();
        // register tables first so that material / texture indices equal the description's
        for (const TextureRef &t : L.tex) out->flat.texture_index(t);
        for (const MaterialRef &m : L.mat) out->flat.material_index(m);
        L.load_set(0, out->objects, 0);
        L.load_set(1, out->lights.objects, 0);
        if (scene->n_media && !scene->media) throw std::runtime_error("media is NULL");
        for (uint32_t k = 0; k < scene->n_media; k++) {
            const pts_medium &m = scene->media[k];
            if (m.texture < 0 || (size_t)m.texture >= L.tex.size()) throw std::runtime_error("medium: texture index out of range");
            if (!(m.radius > 0.0f) || !(m.density > 0.0f)) throw std::runtime_error("medium: radius and density must be positive");
            out->flat.add_medium(Point3(m.center[0], m.center[1], m.center[2]), m.radius, m.density, L.tex[m.texture]);
        }
        if (scene->n_moving && !scene->moving) throw std::runtime_error("moving is NULL");
        for (uint32_t k = 0; k < scene->n_moving; k++) {
            const pts_moving_sphere &m = scene->moving[k];
            if (m.material < -1 || m.material >= (int32_t)L.mat.size()) throw std::runtime_error("moving sphere: material index out of range");
            out->flat.add_moving_sphere(Point3(m.center0[0], m.center0[1], m.center0[2]), Point3(m.center1[0], m.center1[1], m.center1[2]), m.radius,
                                        m.material >= 0 ? L.mat[m.material] : MaterialRef());
        }
        out->flat.build(out->objects, out->lights);
        const pts_camera &c = *camera;
        out->camera = std::make_unique<Camera>(c.aspect_ratio, c.image_width, c.vfov, Point3(c.look_from[0], c.look_from[1], c.look_from[2]),
                                               Point3(c.look_at[0], c.look_at[1], c.look_at[2]), c.focal_length, c.defocus_angle, c.spp, c.max_depth,
                                               Color(c.background[0], c.background[1], c.background[2]));
        return out.release();
    } catch (const std::exception &e) {
        put_err(e.what());
        return nullptr;
    } catch (...) {
        put_err("unknown error");
        return nullptr;
    }
}

void pth_scene_free(pth_scene *s) { delete s; }
const pt_scene_desc *pth_scene_desc(const pth_scene *s) { return s ? &s->flat.desc() : nullptr; }
const pt_camera_desc *pth_camera_desc(const pth_scene *s) { return s ? &s->camera->desc() : nullptr; }

uint32_t pth_perm(const pth_scene *s, int set, int type, uint32_t *out) {
    if (!s) return 0;
    const PrimitiveBuffer &pb = set == 0 ? s->objects : s->lights.objects;
    const std::vector<uint32_t> *p = nullptr;
    switch (type) {
        case PT_PRIM_TRI: p = &pb.triangles.perm; break;
        case PT_PRIM_SPHERE: p = &pb.spheres.perm; break;
        case PT_PRIM_QUAD: p = &pb.quads.perm; break;
        case PT_PRIM_INSTANCE: p = &pb.instances.perm; break;
        default: return 0;
    }
    if (out && !p->empty()) std::memcpy(out, p->data(), p->size() * sizeof(uint32_t));
    return (uint32_t)p->size();
}

int pth_camera_render(const pth_scene *s, pt_context *ctx, uint64_t seed, uint8_t *rgb8) {
    if (!s || !ctx || !rgb8) return PT_ERR_INVALID;
    int rc = pt_upload_scene(ctx, &s->flat.desc());
    if (rc) return rc;
    if ((rc = pt_set_camera(ctx, &s->camera->desc()))) return rc;
    const uint32_t total = s->camera->get_sqrt_spp() * s->camera->get_sqrt_spp();
    if ((rc = pt_render(ctx, 0, total, seed))) return rc;
    return pt_read_rgb8(ctx, rgb8);
}

}  // extern "C"
