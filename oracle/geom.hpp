// TEST INFRASTRUCTURE — CPU oracle (see na.hpp).
//
// Ray, HitRecord, BBox, the three primitives, the binned-SAH builder, the flat
// node array and its traversal, PrimitiveBuffer / LightBuffer / Instance.
// Each function names the reference lines it restates (paths relative to the
// reference checkout).  parity unpinned: the reference has no tests for any of it.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "na.hpp"
#include "rng.hpp"

namespace oracle {

// Per-thread counters of work done by the reference-order traversal: the numbers
// the bytes-per-ray model is built from.
struct Counters {
    uint64_t nodes = 0, tris = 0, spheres = 0, quads = 0, rays = 0;
    uint64_t max_stack = 0;            // deepest traversal stack seen
    uint64_t stack_hist[65] = {0};     // per ray: histogram of its deepest stack
    // by bounce index (0 = camera ray): rays, AABB tests, and the largest AABB-test count of one ray
    uint64_t depth_rays[64] = {0}, depth_nodes[64] = {0}, depth_max_nodes[64] = {0};
};
inline thread_local Counters *g_ctr = nullptr;

// src/types/ray.rs:4-38
struct Ray {
    V3 origin, direction;
    float t_min, t_max;
    static Ray make(V3 o, V3 d) { return Ray{o, d, 0.001f, std::numeric_limits<float>::infinity()}; }
    V3 at(float t) const { return origin + t * direction; }
};

enum PrimType : int32_t { PRIM_NONE = -1, PRIM_TRI = 0, PRIM_SPHERE = 1, PRIM_QUAD = 2, PRIM_INSTANCE = 3 };

// src/objects.rs:14-91
struct HitRecord {
    V3 p, normal;
    float t;
    bool front_face;
    int32_t material;  // -1 = None
    float u, v;
    // identification of what was hit, for the first-hit comparison
    int32_t prim_type = PRIM_NONE, prim_id = -1, inst_id = -1;
    float bary_u = 0.0f, bary_v = 0.0f;  // triangle barycentrics / quad alpha,beta

    static HitRecord make(const Ray &ray, V3 p, V3 normal, float t, int32_t material, float u, float v) {
        HitRecord r;
        r.p = p;
        r.t = t;
        r.material = material;
        r.u = u;
        r.v = v;
        // set_face_normal, src/objects.rs:82-90
        r.front_face = dot(ray.direction, normal) < 0.0f;
        r.normal = r.front_face ? normal : -normal;
        return r;
    }
};

// src/bvh.rs:33-281
struct BBox {
    V3 min, max;
    static BBox make(V3 a, V3 b) {  // :42-49
        BBox r;
        r.min = v3(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z));
        r.max = v3(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z));
        pad(r, 0.0001f);
        return r;
    }
    static void pad(BBox &b, float padding) {  // :51-65
        const float EPS = 0.0001f;
        if (std::fabs(b.max.x - b.min.x) < EPS) { b.min.x -= padding; b.max.x += padding; }
        if (std::fabs(b.max.y - b.min.y) < EPS) { b.min.y -= padding; b.max.y += padding; }
        if (std::fabs(b.max.z - b.min.z) < EPS) { b.min.z -= padding; b.max.z += padding; }
    }
    static BBox empty() {  // :68-73
        float inf = std::numeric_limits<float>::infinity();
        return BBox{v3(inf, inf, inf), v3(-inf, -inf, -inf)};
    }
    static BBox point(V3 p) { return BBox{p, p}; }
    V3 centroid() const { return center(min, max); }  // :84-86
    BBox merge(const BBox &o) const {                  // :89-101
        return BBox{v3(rmin(min.x, o.min.x), rmin(min.y, o.min.y), rmin(min.z, o.min.z)),
                    v3(rmax(max.x, o.max.x), rmax(max.y, o.max.y), rmax(max.z, o.max.z))};
    }
    BBox enclose(V3 p) const {  // :104-116
        return BBox{v3(rmin(min.x, p.x), rmin(min.y, p.y), rmin(min.z, p.z)),
                    v3(rmax(max.x, p.x), rmax(max.y, p.y), rmax(max.z, p.z))};
    }
    float surface_area() const {  // :140-143
        V3 e = max - min;
        return std::fabs(dot(v3(e.x, e.x, e.y), v3(e.y, e.z, e.z))) * 2.0f;
    }
    int longest_axis() const {  // :237-246
        V3 e = max - min;
        if (e.x > e.y && e.x > e.z) return 0;
        if (e.y > e.z) return 1;
        return 2;
    }
    V3 offset(V3 p) const {  // :248-261
        V3 o = p - min;
        if (max.x > min.x) o.x /= max.x - min.x;
        if (max.y > min.y) o.y /= max.y - min.y;
        if (max.z > min.z) o.z /= max.z - min.z;
        return o;
    }
    // :145-182
    bool intersect(const Ray &ray) const {
        if (g_ctr) g_ctr->nodes++;
        float t_min = ray.t_min, t_max = ray.t_max;
        for (int axis = 0; axis < 3; axis++) {
            float lo = min[axis], hi = max[axis];
            float adinv = 1.0f / ray.direction[axis];
            float t0 = (lo - ray.origin[axis]) * adinv;
            float t1 = (hi - ray.origin[axis]) * adinv;
            if (t0 < t1) {
                if (t0 > t_min) t_min = t0;
                if (t1 < t_max) t_max = t1;
            } else {
                if (t1 > t_min) t_min = t1;
                if (t0 < t_max) t_max = t0;
            }
            if (t_max < t_min) return false;
        }
        return true;
    }
    BBox transform(const M4 &m) const {  // :263-280
        V3 a1 = transform_point(m, min), a2 = transform_point(m, max);
        V3 b1 = transform_point(m, v3(max.x, min.y, min.z)), b2 = transform_point(m, v3(min.x, max.y, max.z));
        V3 c1 = transform_point(m, v3(max.x, max.y, min.z)), c2 = transform_point(m, v3(min.x, min.y, max.z));
        V3 d1 = transform_point(m, v3(min.x, max.y, min.z)), d2 = transform_point(m, v3(max.x, min.y, max.z));
        return BBox::make(a1, a2).merge(BBox::make(b1, b2)).merge(BBox::make(c1, c2)).merge(BBox::make(d1, d2));
    }
};

// ---------------------------------------------------------------------------
// src/objects/tri_mesh.rs:14-278
struct Triangle {
    float data[8][3];  // row = attribute (px,py,pz,nx,ny,nz,tu,tv), column = vertex (:47-72)
    int32_t material;

    static Triangle make(const V3 v[3], const V3 *n, const float (*uv)[2], int32_t material) {  // :29-75
        Triangle t;
        V3 nn[3];
        if (n) {
            nn[0] = n[0]; nn[1] = n[1]; nn[2] = n[2];
        } else {
            V3 g = normalize(cross(v[1] - v[0], v[2] - v[0]));  // get_normal :17-19
            nn[0] = nn[1] = nn[2] = g;
        }
        for (int c = 0; c < 3; c++) {
            t.data[0][c] = v[c].x; t.data[1][c] = v[c].y; t.data[2][c] = v[c].z;
            t.data[3][c] = nn[c].x; t.data[4][c] = nn[c].y; t.data[5][c] = nn[c].z;
            t.data[6][c] = uv ? uv[c][0] : 0.0f;
            t.data[7][c] = uv ? uv[c][1] : 0.0f;
        }
        t.material = material;
        return t;
    }
    V3 position(int c) const { return v3(data[0][c], data[1][c], data[2][c]); }

    bool hit(Ray &ray, HitRecord &out) const {  // :167-220
        if (g_ctr) g_ctr->tris++;
        V3 v0 = position(0), v1 = position(1), v2 = position(2);
        V3 e1 = v1 - v0, e2 = v2 - v0;
        V3 d = ray.direction;
        V3 e1_x_d = cross(e1, d);
        float triple = dot(e1_x_d, e2);
        if (triple == 0.0f) return false;
        float inv_triple = 1.0f / triple;
        V3 s = ray.origin - v0;
        V3 e2_x_s = cross(e2, s);
        float u = dot(e2_x_s, d) * inv_triple;
        if (u < 0.0f || u > 1.0f) return false;
        float v = dot(e1_x_d, s) * inv_triple;
        if (v < 0.0f || u + v > 1.0f) return false;
        float t = dot(e2_x_s, e1) * inv_triple;
        if (!(ray.t_min <= t && t <= ray.t_max)) return false;
        // bary_interpolate (:133-137): 8x3 matrix times (1-u-v, u, v), column accumulation
        float w0 = 1.0f - u - v, w1 = u, w2 = v;
        float a[8];
        for (int r = 0; r < 8; r++) a[r] = data[r][0] * w0;
        for (int r = 0; r < 8; r++) a[r] = data[r][1] * w1 + a[r];
        for (int r = 0; r < 8; r++) a[r] = data[r][2] * w2 + a[r];
        V3 normal = normalize(v3(a[3], a[4], a[5]));
        V3 position = v3(a[0], a[1], a[2]);
        ray.t_max = t;
        out = HitRecord::make(ray, position, normal, t, material, a[6], a[7]);
        out.prim_type = PRIM_TRI;
        out.bary_u = u;
        out.bary_v = v;
        return true;
    }
    BBox bbox() const {  // :226-250
        V3 p0 = position(0), p1 = position(1), p2 = position(2);
        V3 lo = v3(rmin(p0.x, rmin(p1.x, p2.x)), rmin(p0.y, rmin(p1.y, p2.y)), rmin(p0.z, rmin(p1.z, p2.z)));
        V3 hi = v3(rmax(p0.x, rmax(p1.x, p2.x)), rmax(p0.y, rmax(p1.y, p2.y)), rmax(p0.z, rmax(p1.z, p2.z)));
        return BBox::make(lo, hi);
    }
    float pdf(const Ray &ray) const {  // :252-261
        Ray r = ray;
        HitRecord rec;
        if (!hit(r, rec)) return 0.0f;
        V3 u = position(1) - position(0), v = position(2) - position(0);
        return 2.0f / norm(cross(u, v));
    }
    // Not in the reference: the density over directions that `sample` really has (the reference's pdf above
    // is 1 / area, a density over the triangle); only the next-event estimator uses it.
    float pdf_solid_angle(const Ray &ray) const {
        Ray r = ray;
        HitRecord rec;
        if (!hit(r, rec)) return 0.0f;
        V3 n = cross(position(1) - position(0), position(2) - position(0));
        float nn = norm(n);
        float distance_squared = rec.t * rec.t * norm_squared(ray.direction);
        float cosine = std::fabs(dot(ray.direction, n)) / (norm(ray.direction) * nn);
        return distance_squared / (cosine * (0.5f * nn));
    }
    V3 sample(Rng &rng, V3 origin) const {  // :263-277
        float s = rng.range(0.0f, 1.0f), t = rng.range(0.0f, 1.0f);
        V3 u = position(1) - position(0), v = position(2) - position(0);
        V3 offset = (s + t <= 1.0f) ? (s * u + t * v) : ((1.0f - s) * u + (1.0f - t) * v);
        return (position(0) - origin) + offset;
    }
};

// ---------------------------------------------------------------------------
// src/types/onb.rs:9-44
struct ONB {
    M3 m;
    static ONB make(V3 n) {
        V3 v = normalize(n);
        V3 a = (std::fabs(v.x) > 0.9f) ? v3(0, 1, 0) : v3(1, 0, 0);
        V3 w = normalize(cross(v, a));
        V3 u = cross(v, w);
        return ONB{M3{u, v, w}};
    }
    V3 to_local(V3 x) const { return mul_transpose(m, x); }
    V3 to_world(V3 x) const { return mul(m, x); }
};

// src/objects/sphere.rs:9-129
struct Sphere {
    V3 center;
    float radius;
    int32_t material;
    BBox box;
    static Sphere make(V3 c, float r, int32_t material) {  // :19-27
        V3 rv = v3(r, r, r);
        return Sphere{c, r, material, BBox::make(c - rv, c + rv)};
    }
    static void get_uv(V3 p, float &u, float &v) {  // :29-36
        float theta = std::acos(-p.y);
        float phi = std::atan2(-p.z, p.x) + PI_F;
        u = phi / (2.0f * PI_F);
        v = theta / PI_F;
    }
    bool hit(Ray &ray, HitRecord &out) const {  // :57-99
        if (g_ctr) g_ctr->spheres++;
        V3 oc = center - ray.origin;
        float a = norm_squared(ray.direction);
        float h = dot(ray.direction, oc);
        float c = norm_squared(oc) - radius * radius;
        float discriminant = h * h - a * c;
        if (discriminant < 0.0f) return false;
        float sqrt_disc = std::sqrt(discriminant);
        float root = (h - sqrt_disc) / a;
        if (root < ray.t_min || ray.t_max < root) {
            root = (h + sqrt_disc) / a;
            if (root < ray.t_min || ray.t_max < root) return false;
        }
        V3 outward = (ray.at(root) - center) / radius;
        float u, v;
        get_uv(outward, u, v);
        ray.t_max = root;
        out = HitRecord::make(ray, ray.at(root), outward, root, material, u, v);
        out.prim_type = PRIM_SPHERE;
        return true;
    }
    BBox bbox() const { return box; }
    float pdf(const Ray &ray) const {  // :109-120
        Ray r = ray;
        HitRecord rec;
        if (!hit(r, rec)) return 0.0f;
        float cos_theta_max = std::sqrt(1.0f - radius * radius / norm_squared(center - ray.origin));
        float solid_angle = 2.0f * PI_F * (1.0f - cos_theta_max);
        return 1.0f / solid_angle;
    }
    V3 random_to_sphere(Rng &rng, float rad, float distance_squared) const {  // :38-53
        float r1 = rng.range(0.0f, 1.0f), r2 = rng.range(0.0f, 1.0f);
        float z = 1.0f + r2 * (std::sqrt(1.0f - radius * radius / distance_squared) - 1.0f);
        float phi = 2.0f * PI_F * r1;
        float x = std::cos(phi) * std::sqrt(1.0f - z * z);
        float y = std::sin(phi) * std::sqrt(1.0f - z * z);
        return v3(x, y, z) * rad;
    }
    V3 sample(Rng &rng, V3 origin) const {  // :122-128
        V3 direction = center - origin;
        float distance_squared = norm_squared(direction);
        ONB uvw = ONB::make(direction);
        return uvw.to_world(random_to_sphere(rng, radius, distance_squared));
    }
    // Not in the reference: `sample` puts the cone of random_to_sphere around the basis' third axis, while
    // OrthonormalBasis::new puts the direction it is given on the second (src/types/onb.rs), so the reference's
    // samples point sideways and do not have the density `pdf` reports.  This one swaps the two components;
    // only the next-event estimator uses it.
    V3 sample_towards(Rng &rng, V3 origin) const {
        V3 direction = center - origin;
        float distance_squared = norm_squared(direction);
        ONB uvw = ONB::make(direction);
        V3 c = random_to_sphere(rng, radius, distance_squared);
        return uvw.to_world(v3(c.x, c.z, c.y));
    }
};

// src/objects/quad_mesh.rs:10-180
struct Quad {
    V3 origin, u, v, normal, w;
    float d, area;
    int32_t material;
    BBox box;
    static Quad make(V3 origin, V3 u, V3 v, int32_t material) {  // :25-52
        Quad q;
        BBox d1 = BBox::make(origin, origin + u + v);
        BBox d2 = BBox::make(origin + u, origin + v);
        q.box = d1.merge(d2);
        V3 n = cross(u, v);
        q.normal = normalize(n);
        q.d = dot(q.normal, origin);
        q.w = n / dot(n, n);
        q.area = norm(n);
        q.origin = origin;
        q.u = u;
        q.v = v;
        q.material = material;
        return q;
    }
    bool hit(Ray &ray, HitRecord &out) const {  // :126-158
        if (g_ctr) g_ctr->quads++;
        float denom = dot(normal, ray.direction);
        if (std::fabs(denom) < 1e-6f) return false;
        float t = (d - dot(normal, ray.origin)) / denom;
        if (t < ray.t_min || t > ray.t_max) return false;
        V3 intersection = ray.at(t);
        V3 p = intersection - origin;
        float alpha = dot(w, cross(p, v));
        float beta = dot(w, cross(u, p));
        if (!(0.0f <= alpha && alpha <= 1.0f) || !(0.0f <= beta && beta <= 1.0f)) return false;
        ray.t_max = t;
        out = HitRecord::make(ray, intersection, normal, t, material, alpha, beta);
        out.prim_type = PRIM_QUAD;
        out.bary_u = alpha;
        out.bary_v = beta;
        return true;
    }
    BBox bbox() const { return box; }
    float pdf(const Ray &ray) const {  // :165-175
        Ray r = ray;
        HitRecord rec;
        if (!hit(r, rec)) return 0.0f;
        float distance_squared = rec.t * rec.t * norm_squared(ray.direction);
        float cosine = std::fabs(dot(ray.direction, rec.normal)) / norm(ray.direction);
        return distance_squared / (cosine * area);
    }
    V3 sample(Rng &rng, V3 from) const {  // :176-179
        float a = rng.range(0.0f, 1.0f);
        float b = rng.range(0.0f, 1.0f);
        V3 p = origin + u * a + v * b;
        return p - from;
    }
};

// ---------------------------------------------------------------------------
// src/bvh.rs:667-703.  One flat node; the left child of an interior node is the
// next node in the array.
struct FlatNode {
    BBox bbox;
    uint32_t is_leaf;
    uint32_t a;  // interior: right_child_offset; leaf: offset
    uint32_t b;  // interior: axis;               leaf: size (u16 in the reference)
};

struct Instance;

template <class T>
struct InnerBuffer {  // src/objects.rs:105-128
    std::vector<T> buffer;
    std::vector<FlatNode> bvh;
    bool has_bvh = false;
    BBox bbox = BBox::empty();
    // original insertion index of buffer[i] after the builder permuted it
    std::vector<uint32_t> perm;

    void push(const T &p) {
        bbox = bbox.merge(p.bbox());
        perm.push_back((uint32_t)buffer.size());
        buffer.push_back(p);
    }
    size_t len() const { return buffer.size(); }
    // as_instances: the accepted element's index goes to inst_id and the inner
    // primitive's own (prim_type, prim_id) is kept.
    bool traverse(Ray &ray, HitRecord &out, bool as_instances = false) const;
};

// src/bvh.rs:11-31
template <class T, class F>
size_t partition_mut(std::vector<T> &items, std::vector<uint32_t> &perm, size_t start, size_t end, F cmp) {
    size_t i = start, j = end - 1;
    while (i < j) {
        if (!cmp(items[j])) { j -= 1; continue; }
        if (cmp(items[i])) { i += 1; continue; }
        std::swap(items[i], items[j]);
        std::swap(perm[i], perm[j]);
    }
    return i + 1;
}

// Build-time tree (src/bvh.rs:518-530)
struct BuildNode {
    bool leaf;
    std::unique_ptr<BuildNode> left, right;
    BBox bbox;
    uint8_t axis;
    size_t start, size;
};

// src/bvh.rs:289-516 with SplitMethod::SAH (the only method build_bvh selects,
// src/objects.rs:172-187) and max_leaf_size 3 (:301).
template <class T>
struct Builder {
    static constexpr size_t MAX_LEAF = 3;

    static bool build(InnerBuffer<T> &objs) {  // :295-311
        auto root = build_rec(objs, 0, objs.buffer.size(), 0);
        std::vector<FlatNode> nodes;
        if (!flatten(*root, nodes, 0)) return false;
        objs.bvh = std::move(nodes);
        objs.has_bvh = true;
        return true;
    }

    static std::unique_ptr<BuildNode> leaf(size_t start, size_t size, BBox b) {
        auto n = std::make_unique<BuildNode>();
        n->leaf = true; n->bbox = b; n->start = start; n->size = size; n->axis = 0;
        return n;
    }
    static std::unique_ptr<BuildNode> interior(std::unique_ptr<BuildNode> l, std::unique_ptr<BuildNode> r, BBox b, uint8_t axis) {
        auto n = std::make_unique<BuildNode>();
        n->leaf = false; n->left = std::move(l); n->right = std::move(r); n->bbox = b; n->axis = axis;
        n->start = n->size = 0;
        return n;
    }

    static std::unique_ptr<BuildNode> build_rec(InnerBuffer<T> &objs, size_t start, size_t end, size_t depth) {  // :313-468
        assert(start < end);
        std::vector<T> &buf = objs.buffer;
        BBox bbox = buf[start].bbox();
        for (size_t i = start; i < end; i++) bbox = bbox.merge(buf[i].bbox());
        size_t size = end - start;
        if (size == 1) return leaf(start, size, bbox);
        int axis = bbox.longest_axis();
        BBox cb = BBox::point(buf[start].bbox().centroid());
        for (size_t i = start; i < end; i++) cb = cb.enclose(buf[i].bbox().centroid());
        if (cb.max[axis] - cb.min[axis] <= 0.0001f) return leaf(start, size, bbox);

        if (size <= 2) {  // :365-372  (slice::sort_by is a stable sort)
            // size == 2 here: the stable sort swaps iff the second centroid is strictly smaller
            if (buf[start + 1].bbox().centroid()[axis] < buf[start].bbox().centroid()[axis]) {
                std::swap(buf[start], buf[start + 1]);
                std::swap(objs.perm[start], objs.perm[start + 1]);
            }
            size_t mid = start + size / 2;
            auto l = build_rec(objs, start, mid, depth + 1);
            auto r = build_rec(objs, mid, end, depth + 1);
            return interior(std::move(l), std::move(r), bbox, (uint8_t)axis);
        }

        constexpr int NB = 12;  // :374
        uint32_t count[NB] = {0};
        bool has[NB] = {false};
        BBox bb[NB];
        auto bucket_of = [&](const T &o) {
            float off = cb.offset(o.bbox().centroid())[axis];
            // `(12.0 * off) as usize`: saturating, NaN -> 0
            float f = (float)NB * off;
            size_t b;
            if (!(f > 0.0f)) b = 0;
            else if (f >= 1.8e19f) b = (size_t)-1;
            else b = (size_t)f;
            return std::min<size_t>(b, NB - 1);
        };
        for (size_t i = start; i < end; i++) {  // :378-397
            size_t b = bucket_of(buf[i]);
            count[b] += 1;
            if (has[b]) bb[b] = bb[b].merge(buf[i].bbox());
            else { bb[b] = buf[i].bbox(); has[b] = true; }
        }
        float cost[NB - 1] = {0};
        {  // :399-416
            bool hl = false; BBox lb{}; uint32_t lc = 0;
            for (int i = 0; i < NB - 1; i++) {
                if (hl && has[i]) lb = lb.merge(bb[i]);
                else if (!hl && has[i]) { lb = bb[i]; hl = true; }
                lc += count[i];
                cost[i] += (float)lc * (hl ? lb.surface_area() : 0.0f);
            }
        }
        {  // :418-434
            bool hr = false; BBox rb{}; uint32_t rc = 0;
            for (int i = NB - 1; i >= 1; i--) {
                if (hr && has[i]) rb = rb.merge(bb[i]);
                else if (!hr && has[i]) { rb = bb[i]; hr = true; }
                rc += count[i];
                cost[i - 1] += (float)rc * (hr ? rb.surface_area() : 0.0f);
            }
        }
        float min_cost = std::numeric_limits<float>::infinity();  // :436-443
        int split = NB / 2;
        for (int i = 0; i < NB - 1; i++)
            if (cost[i] < min_cost) { min_cost = cost[i]; split = i; }
        min_cost = 0.5f + (((float)size * min_cost) / bbox.surface_area());  // :445
        float leaf_cost = (float)size;
        if ((size > MAX_LEAF) || (min_cost < leaf_cost)) {  // :447-456
            size_t mid = partition_mut(buf, objs.perm, start, end,
                                       [&](const T &o) { return bucket_of(o) <= (size_t)split; });
            auto l = build_rec(objs, start, mid, depth + 1);
            auto r = build_rec(objs, mid, end, depth + 1);
            return interior(std::move(l), std::move(r), bbox, (uint8_t)axis);
        }
        return leaf(start, size, bbox);
    }

    static bool flatten(const BuildNode &n, std::vector<FlatNode> &nodes, uint32_t depth) {  // :478-515
        if (depth > 63) return false;
        if (!n.leaf) {
            size_t off = nodes.size();
            nodes.push_back(FlatNode{n.bbox, 0, (uint32_t)(off + 1), n.axis});
            if (!flatten(*n.left, nodes, depth + 1)) return false;
            size_t right = nodes.size();
            if (!flatten(*n.right, nodes, depth + 1)) return false;
            nodes[off].a = (uint32_t)right;
        } else {
            nodes.push_back(FlatNode{n.bbox, 1, (uint32_t)n.start, (uint32_t)(uint16_t)n.size});  // `size as u16`
        }
        return true;
    }
};

// src/bvh.rs:725-797
template <class T>
bool InnerBuffer<T>::traverse(Ray &ray, HitRecord &out, bool as_instances) const {
    bool any = false;
    HitRecord h;
    if (!has_bvh) {  // :726-737
        for (size_t i = 0; i < buffer.size(); i++)
            if (buffer[i].hit(ray, h)) { out = h; (as_instances ? out.inst_id : out.prim_id) = (int32_t)i; any = true; }
        return any;
    }
    size_t cur = 0, top = 0, deepest = 0;
    size_t stack[64];
    bool neg[3] = {ray.direction.x < 0.0f, ray.direction.y < 0.0f, ray.direction.z < 0.0f};
    for (;;) {
        const FlatNode &n = bvh[cur];
        if (n.bbox.intersect(ray)) {
            if (!n.is_leaf) {
                if (neg[n.b]) { stack[top++] = cur + 1; cur = n.a; }
                else { stack[top++] = n.a; cur += 1; }
                if (top > deepest) deepest = top;
            } else {
                for (size_t i = n.a; i < (size_t)n.a + n.b; i++)
                    if (buffer[i].hit(ray, h)) { out = h; (as_instances ? out.inst_id : out.prim_id) = (int32_t)i; any = true; }
                if (top == 0) break;
                cur = stack[--top];
            }
        } else {
            if (top == 0) break;
            cur = stack[--top];
        }
    }
    if (g_ctr) {
        if (deepest > g_ctr->max_stack) g_ctr->max_stack = deepest;
        g_ctr->stack_hist[deepest]++;
    }
    return any;
}

// src/objects.rs:130-206
struct PrimitiveBuffer {
    InnerBuffer<Triangle> triangles;
    InnerBuffer<Sphere> spheres;
    InnerBuffer<Quad> quads;
    InnerBuffer<Instance> instances;
    BBox bbox = BBox::empty();

    void add_triangle(const Triangle &t) { bbox = bbox.merge(t.bbox()); triangles.push(t); }
    void add_sphere(const Sphere &s) { bbox = bbox.merge(s.bbox()); spheres.push(s); }
    void add_quad(const Quad &q) { bbox = bbox.merge(q.bbox()); quads.push(q); }
    void add_instance(const Instance &i);
    bool build_bvh();
    bool hit(Ray &ray, HitRecord &out) const;
    size_t len() const;
};

// src/objects.rs:281-410
struct Instance {
    std::shared_ptr<PrimitiveBuffer> obj;
    M4 transform, inverse;

    static Instance from_obj(std::shared_ptr<PrimitiveBuffer> o) { return Instance{std::move(o), m4_identity(), m4_identity()}; }
    void rotate(float angle, V3 axis) { transform = mul(m4_rotation(angle * axis), transform); inverse = mul(inverse, m4_rotation(-angle * axis)); }
    void rotate_x(float a) { transform = mul(m4_rotation(v3(a, 0, 0)), transform); inverse = mul(inverse, m4_rotation(v3(-a, 0, 0))); }
    void rotate_y(float a) { transform = mul(m4_rotation(v3(0, a, 0)), transform); inverse = mul(inverse, m4_rotation(v3(0, -a, 0))); }
    void rotate_z(float a) { transform = mul(m4_rotation(v3(0, 0, a)), transform); inverse = mul(inverse, m4_rotation(v3(0, 0, -a))); }
    void translate(V3 s) { transform = mul(m4_translation(s), transform); inverse = mul(inverse, m4_translation(-s)); }
    void scale(V3 s) { transform = mul(m4_scaling(s), transform); inverse = mul(inverse, m4_scaling(v3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z))); }

    bool hit(Ray &ray, HitRecord &out) const {  // :384-401
        Ray local{transform_point(inverse, ray.origin), transform_vector(inverse, ray.direction), ray.t_min, ray.t_max};
        HitRecord rec;
        if (!obj->hit(local, rec)) return false;
        rec.p = transform_point(transform, rec.p);
        rec.normal = transform_vector(transform, rec.normal);
        ray.t_max = rec.t;
        out = rec;
        return true;
    }
    BBox bbox() const { return obj->bbox.transform(transform); }  // :407-409
};

inline void PrimitiveBuffer::add_instance(const Instance &i) { bbox = bbox.merge(i.bbox()); instances.push(i); }
inline size_t PrimitiveBuffer::len() const { return triangles.len() + spheres.len() + quads.len() + instances.len(); }
inline bool PrimitiveBuffer::build_bvh() {  // :169-189
    bool ok = true;
    if (triangles.len() > 0) ok = Builder<Triangle>::build(triangles) && ok;
    if (spheres.len() > 0) ok = Builder<Sphere>::build(spheres) && ok;
    if (quads.len() > 0) ok = Builder<Quad>::build(quads) && ok;
    if (instances.len() > 0) ok = Builder<Instance>::build(instances) && ok;
    return ok;
}
inline bool PrimitiveBuffer::hit(Ray &ray, HitRecord &out) const {  // :191-197
    bool any = false;
    HitRecord h;
    if (triangles.traverse(ray, h)) { out = h; any = true; }
    if (spheres.traverse(ray, h)) { out = h; any = true; }
    if (quads.traverse(ray, h)) { out = h; any = true; }
    if (instances.traverse(ray, h, true)) { out = h; any = true; }
    return any;
}

// src/objects.rs:208-279
struct LightBuffer {
    PrimitiveBuffer objects;

    V3 sample(Rng &rng, V3 origin) const {  // :235-257
        size_t n = objects.len();
        size_t i = (size_t)rng.below(n);
        size_t nt = objects.triangles.len(), ns = objects.spheres.len(), nq = objects.quads.len();
        if (i < nt) return objects.triangles.buffer[i].sample(rng, origin);
        if (i < nt + ns) return objects.spheres.buffer[i - nt].sample(rng, origin);
        if (i < nt + ns + nq) return objects.quads.buffer[i - nt - ns].sample(rng, origin);
        return v3(0, 0, 0);  // Instance uses the trait default (:100-102)
    }
    // the same with Sphere::sample_towards.  *picked: the index of the light that was drawn
    V3 sample_nee(Rng &rng, V3 origin, size_t *picked = nullptr) const {
        size_t n = objects.len();
        size_t i = (size_t)rng.below(n);
        size_t nt = objects.triangles.len(), ns = objects.spheres.len(), nq = objects.quads.len();
        if (picked) *picked = i;
        if (i < nt) return objects.triangles.buffer[i].sample(rng, origin);
        if (i < nt + ns) return objects.spheres.buffer[i - nt].sample_towards(rng, origin);
        if (i < nt + ns + nq) return objects.quads.buffer[i - nt - ns].sample(rng, origin);
        return v3(0, 0, 0);
    }
    float pdf(const Ray &ray) const {  // :259-278
        float pdf = 0.0f;
        for (const auto &t : objects.triangles.buffer) pdf += t.pdf(ray);
        for (const auto &s : objects.spheres.buffer) pdf += s.pdf(ray);
        for (const auto &q : objects.quads.buffer) pdf += q.pdf(ray);
        for (size_t i = 0; i < objects.instances.len(); i++) pdf += 0.0f;  // trait default (:97-99)
        return pdf / (float)objects.len();
    }
    // Light i's own density over directions for the ray (Triangle::pdf_solid_angle, Sphere::pdf, Quad::pdf; 0 when the
    // ray misses it or for an instance), and where the ray meets that light.
    float term_nee(size_t i, const Ray &ray, HitRecord &rec) const {
        size_t nt = objects.triangles.len(), ns = objects.spheres.len(), nq = objects.quads.len();
        Ray r = ray;
        if (i < nt) return objects.triangles.buffer[i].hit(r, rec) ? objects.triangles.buffer[i].pdf_solid_angle(ray) : 0.0f;
        if (i < nt + ns) return objects.spheres.buffer[i - nt].hit(r, rec) ? objects.spheres.buffer[i - nt].pdf(ray) : 0.0f;
        if (i < nt + ns + nq) return objects.quads.buffer[i - nt - ns].hit(r, rec) ? objects.quads.buffer[i - nt - ns].pdf(ray) : 0.0f;
        return 0.0f;
    }
    static bool same_distance(float a, float b) { return std::fabs(a - b) <= 1.0e-4f * rmax(1.0f, std::fabs(b)); }
    // The density with which light sampling yields the point the ray reaches at distance t_hit: the terms of the
    // lights the ray meets at that distance (the emitter that was hit, as a light), over the number of lights.
    float pdf_nee_at(const Ray &ray, float t_hit) const {
        float pdf = 0.0f;
        HitRecord rec;
        for (size_t i = 0; i < objects.len(); i++) {
            float term = term_nee(i, ray, rec);
            if (term != 0.0f && same_distance(rec.t, t_hit)) pdf += term;
        }
        return pdf / (float)objects.len();
    }
    float pdf_nee(const Ray &ray) const {  // the same with Triangle::pdf_solid_angle
        float pdf = 0.0f;
        for (const auto &t : objects.triangles.buffer) pdf += t.pdf_solid_angle(ray);
        for (const auto &s : objects.spheres.buffer) pdf += s.pdf(ray);
        for (const auto &q : objects.quads.buffer) pdf += q.pdf(ray);
        for (size_t i = 0; i < objects.instances.len(); i++) pdf += 0.0f;
        return pdf / (float)objects.len();
    }
};

}  // namespace oracle
