// TEST INFRASTRUCTURE — CPU oracle (see na.hpp).
//
// C entry points so that the Python tests and the CPU-baseline leg of the
// benchmark can drive the restatement through ctypes.  Nothing here is linked
// into, or called by, the shipped libraries.
#include <cstring>
#include <thread>

#include "../include/ptcore.h"
#include "render.hpp"

using namespace oracle;

struct pto_scene {
    Scene *sc;
};

static const PrimitiveBuffer *pick_set(const pto_scene *s, int set) { return set == 0 ? &s->sc->objects : &s->sc->lights.objects; }

extern "C" {

pto_scene *pto_load(const pts_scene *scene, const pts_camera *cam, char *err, size_t errlen) {
    std::string e;
    Scene *sc = load_scene(*scene, *cam, e);
    if (!sc) {
        if (err && errlen) { std::strncpy(err, e.c_str(), errlen - 1); err[errlen - 1] = 0; }
        return nullptr;
    }
    return new pto_scene{sc};
}
void pto_free(pto_scene *s) {
    if (!s) return;
    delete s->sc;
    delete s;
}
void pto_set_bsdf_probability(pto_scene *s, float p) { s->sc->bsdf_sample_probability = p; }

// [0]=width [1]=height [2..4]=center [5..7]=pixel00 [8..10]=pixel_du [11..13]=pixel_dv
// [14..16]=defocus_disk_u [17..19]=defocus_disk_v [20]=defocus_angle [21]=sqrt_spp
// [22]=inv_sqrt_spp [23]=max_depth [24..26]=background
void pto_camera_constants(const pto_scene *s, float *o) {
    const Camera &c = s->sc->camera;
    o[0] = (float)c.image_width; o[1] = (float)c.image_height;
    o[2] = c.center.x; o[3] = c.center.y; o[4] = c.center.z;
    o[5] = c.pixel00.x; o[6] = c.pixel00.y; o[7] = c.pixel00.z;
    o[8] = c.pixel_du.x; o[9] = c.pixel_du.y; o[10] = c.pixel_du.z;
    o[11] = c.pixel_dv.x; o[12] = c.pixel_dv.y; o[13] = c.pixel_dv.z;
    o[14] = c.defocus_disk_u.x; o[15] = c.defocus_disk_u.y; o[16] = c.defocus_disk_u.z;
    o[17] = c.defocus_disk_v.x; o[18] = c.defocus_disk_v.y; o[19] = c.defocus_disk_v.z;
    o[20] = c.defocus_angle; o[21] = (float)c.sqrt_spp; o[22] = c.inv_sqrt_spp; o[23] = (float)c.max_depth;
    o[24] = c.background.x; o[25] = c.background.y; o[26] = c.background.z;
}

uint32_t pto_buffer_len(const pto_scene *s, int set, int type) {
    const PrimitiveBuffer *b = pick_set(s, set);
    switch (type) {
        case PRIM_TRI: return (uint32_t)b->triangles.len();
        case PRIM_SPHERE: return (uint32_t)b->spheres.len();
        case PRIM_QUAD: return (uint32_t)b->quads.len();
        case PRIM_INSTANCE: return (uint32_t)b->instances.len();
    }
    return 0;
}
static const std::vector<FlatNode> *pick_nodes(const pto_scene *s, int set, int type, const std::vector<uint32_t> **perm) {
    const PrimitiveBuffer *b = pick_set(s, set);
    switch (type) {
        case PRIM_TRI: *perm = &b->triangles.perm; return b->triangles.has_bvh ? &b->triangles.bvh : nullptr;
        case PRIM_SPHERE: *perm = &b->spheres.perm; return b->spheres.has_bvh ? &b->spheres.bvh : nullptr;
        case PRIM_QUAD: *perm = &b->quads.perm; return b->quads.has_bvh ? &b->quads.bvh : nullptr;
        case PRIM_INSTANCE: *perm = &b->instances.perm; return b->instances.has_bvh ? &b->instances.bvh : nullptr;
    }
    *perm = nullptr;
    return nullptr;
}
uint32_t pto_bvh_node_count(const pto_scene *s, int set, int type) {
    const std::vector<uint32_t> *perm;
    const std::vector<FlatNode> *n = pick_nodes(s, set, type, &perm);
    return n ? (uint32_t)n->size() : 0;
}
// bbox: [n][6] = min xyz, max xyz; meta: [n][3] = is_leaf, a, b (see FlatNode)
void pto_bvh_nodes(const pto_scene *s, int set, int type, float *bbox, uint32_t *meta) {
    const std::vector<uint32_t> *perm;
    const std::vector<FlatNode> *n = pick_nodes(s, set, type, &perm);
    if (!n) return;
    for (size_t i = 0; i < n->size(); i++) {
        const FlatNode &f = (*n)[i];
        bbox[6 * i] = f.bbox.min.x; bbox[6 * i + 1] = f.bbox.min.y; bbox[6 * i + 2] = f.bbox.min.z;
        bbox[6 * i + 3] = f.bbox.max.x; bbox[6 * i + 4] = f.bbox.max.y; bbox[6 * i + 5] = f.bbox.max.z;
        meta[3 * i] = f.is_leaf; meta[3 * i + 1] = f.a; meta[3 * i + 2] = f.b;
    }
}
// perm[i] = insertion index (within its set and type) of the primitive now at position i
void pto_perm(const pto_scene *s, int set, int type, uint32_t *out) {
    const std::vector<uint32_t> *perm;
    pick_nodes(s, set, type, &perm);
    if (perm) std::memcpy(out, perm->data(), perm->size() * sizeof(uint32_t));
}
// [n][17] = origin, u, v, normal, w, d, area  (post-build order)
void pto_quads(const pto_scene *s, int set, float *o) {
    const PrimitiveBuffer *b = pick_set(s, set);
    for (size_t i = 0; i < b->quads.len(); i++) {
        const Quad &q = b->quads.buffer[i];
        float *p = o + 17 * i;
        p[0] = q.origin.x; p[1] = q.origin.y; p[2] = q.origin.z;
        p[3] = q.u.x; p[4] = q.u.y; p[5] = q.u.z;
        p[6] = q.v.x; p[7] = q.v.y; p[8] = q.v.z;
        p[9] = q.normal.x; p[10] = q.normal.y; p[11] = q.normal.z;
        p[12] = q.w.x; p[13] = q.w.y; p[14] = q.w.z;
        p[15] = q.d; p[16] = q.area;
    }
}
// [n][24] = the 8x3 attribute matrix, row-major (post-build order)
void pto_triangles(const pto_scene *s, int set, float *o) {
    const PrimitiveBuffer *b = pick_set(s, set);
    for (size_t i = 0; i < b->triangles.len(); i++) std::memcpy(o + 24 * i, b->triangles.buffer[i].data, 24 * sizeof(float));
}
// [n][32] = transform (row-major 4x4), inverse (row-major 4x4)  (post-build order)
void pto_instances(const pto_scene *s, int set, float *o) {
    const PrimitiveBuffer *b = pick_set(s, set);
    for (size_t i = 0; i < b->instances.len(); i++) {
        std::memcpy(o + 32 * i, b->instances.buffer[i].transform.m, 16 * sizeof(float));
        std::memcpy(o + 32 * i + 16, b->instances.buffer[i].inverse.m, 16 * sizeof(float));
    }
}

// rays: [W*H][8] = origin, direction, t_min, t_max
void pto_primary_rays(const pto_scene *s, float *rays) {
    const Camera &c = s->sc->camera;
    for (uint32_t v = 0; v < c.image_height; v++)
        for (uint32_t u = 0; u < c.image_width; u++) {
            Ray r = c.fixed_ray((float)u, (float)v);
            float *o = rays + 8 * ((size_t)v * c.image_width + u);
            o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
            o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
            o[6] = r.t_min; o[7] = r.t_max;
        }
}
// Camera::get_ray for every pixel of one stratum, with the generator keyed as the renderer keys it.
// counters_out[i] (may be null) = number of draws consumed.
void pto_sample_rays(const pto_scene *s, uint64_t seed, uint32_t stratum, float *rays, uint32_t *counters_out) {
    const Camera &c = s->sc->camera;
    uint32_t s_u = stratum / c.sqrt_spp, s_v = stratum % c.sqrt_spp;
    size_t npix = (size_t)c.image_width * c.image_height;
    for (size_t i = 0; i < npix; i++) {
        Rng rng = Rng::for_sample(seed, (uint32_t)i, stratum);
        Ray r = c.get_ray(rng, (float)(i % c.image_width), (float)(i / c.image_width), (float)s_u, (float)s_v);
        float *o = rays + 8 * i;
        o[0] = r.origin.x; o[1] = r.origin.y; o[2] = r.origin.z;
        o[3] = r.direction.x; o[4] = r.direction.y; o[5] = r.direction.z;
        o[6] = r.t_min; o[7] = r.t_max;
        if (counters_out) counters_out[i] = rng.counter;
    }
}

// Closest hit for n rays.  ids: [n][3] = prim_type (-1 = miss), prim_id, inst_id (-1 = none);
// tuv: [n][3] = t, and the triangle barycentrics (u, v) or quad (alpha, beta), 0 for spheres;
// counters (may be null): [5] = nodes, tris, spheres, quads, rays  — accumulated.
void pto_trace(const pto_scene *s, const float *rays, uint64_t n, int32_t *ids, float *tuv, uint64_t *counters, uint32_t threads) {
    unsigned nt = threads ? threads : std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    std::atomic<uint64_t> next{0};
    std::mutex m;
    Counters total;
    auto worker = [&]() {
        Counters local;
        g_ctr = counters ? &local : nullptr;
        for (;;) {
            uint64_t b = next.fetch_add(4096);
            if (b >= n) break;
            uint64_t e = std::min<uint64_t>(n, b + 4096);
            for (uint64_t i = b; i < e; i++) {
                const float *r = rays + 8 * i;
                Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6], r[7]};
                HitRecord rec;
                local.rays++;
                if (s->sc->objects.hit(ray, rec)) {
                    ids[3 * i] = rec.prim_type; ids[3 * i + 1] = rec.prim_id; ids[3 * i + 2] = rec.inst_id;
                    tuv[3 * i] = rec.t; tuv[3 * i + 1] = rec.bary_u; tuv[3 * i + 2] = rec.bary_v;
                } else {
                    ids[3 * i] = -1; ids[3 * i + 1] = -1; ids[3 * i + 2] = -1;
                    tuv[3 * i] = ray.t_max; tuv[3 * i + 1] = 0; tuv[3 * i + 2] = 0;
                }
            }
        }
        g_ctr = nullptr;
        std::lock_guard<std::mutex> l(m);
        total.nodes += local.nodes; total.tris += local.tris; total.spheres += local.spheres;
        total.quads += local.quads; total.rays += local.rays;
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (counters) {
        counters[0] += total.nodes; counters[1] += total.tris; counters[2] += total.spheres;
        counters[3] += total.quads; counters[4] += total.rays;
    }
}


// As pto_trace (single-threaded), then the moving spheres where they are at times[i] (null: time 0), after
// everything else and in index order, as Scene::hit_scene does for a path of that time; a hit on moving sphere
// k is reported as prim_type 4, prim_id k.
void pto_trace_at(const pto_scene *s, const float *rays, const float *times, uint64_t n, int32_t *ids, float *tuv) {
    for (uint64_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6], r[7]};
        HitRecord rec;
        bool any = s->sc->objects.hit(ray, rec);
        int32_t type = any ? rec.prim_type : -1, id = any ? rec.prim_id : -1, inst = any ? rec.inst_id : -1;
        float t = any ? rec.t : ray.t_max, bu = any ? rec.bary_u : 0.0f, bv = any ? rec.bary_v : 0.0f;
        const float time = times ? times[i] : 0.0f;
        for (size_t k = 0; k < s->sc->moving.size(); k++) {
            const pts_moving_sphere &m = s->sc->moving[k];
            V3 c0 = v3(m.center0[0], m.center0[1], m.center0[2]), c1 = v3(m.center1[0], m.center1[1], m.center1[2]);
            Sphere sp = Sphere::make(c0 + time * (c1 - c0), m.radius, m.material);
            HitRecord h;
            if (sp.hit(ray, h)) {
                type = 4; id = (int32_t)k; inst = -1;
                t = h.t; bu = bv = 0.0f;
            }
        }
        ids[3 * i] = type; ids[3 * i + 1] = id; ids[3 * i + 2] = inst;
        tuv[3 * i] = t; tuv[3 * i + 1] = bu; tuv[3 * i + 2] = bv;
    }
}

// As pto_trace, single-threaded, and additionally returns the number of AABB tests and
// primitive tests each ray caused: per_ray[n][2]; stack_hist[65] (may be null) = number of BVH
// traversals whose deepest stack had k entries.
void pto_trace_counts(const pto_scene *s, const float *rays, uint64_t n, int32_t *ids, float *tuv, uint32_t *per_ray, uint64_t *stack_hist) {
    Counters c;
    g_ctr = &c;
    for (uint64_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6], r[7]};
        HitRecord rec;
        uint64_t n0 = c.nodes, p0 = c.tris + c.spheres + c.quads;
        if (s->sc->objects.hit(ray, rec)) {
            ids[3 * i] = rec.prim_type; ids[3 * i + 1] = rec.prim_id; ids[3 * i + 2] = rec.inst_id;
            tuv[3 * i] = rec.t; tuv[3 * i + 1] = rec.bary_u; tuv[3 * i + 2] = rec.bary_v;
        } else {
            ids[3 * i] = -1; ids[3 * i + 1] = -1; ids[3 * i + 2] = -1;
            tuv[3 * i] = ray.t_max; tuv[3 * i + 1] = 0; tuv[3 * i + 2] = 0;
        }
        per_ray[2 * i] = (uint32_t)(c.nodes - n0);
        per_ray[2 * i + 1] = (uint32_t)(c.tris + c.spheres + c.quads - p0);
    }
    g_ctr = nullptr;
    if (stack_hist) for (int k = 0; k < 65; k++) stack_hist[k] = c.stack_hist[k];
}

// Renders strata [begin, end) into accum (W*H*3, added to).  rgb8 / per_sample / rays / samples may be null.
void pto_render(const pto_scene *s, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed, uint32_t threads, int iterative,
                int tonemap_each_pass, float *accum, uint8_t *rgb8, float *per_sample, uint64_t *rays, uint32_t *samples) {
    RenderOptions o{stratum_begin, stratum_end, seed, threads, iterative, tonemap_each_pass};
    render(*s->sc, o, accum, rgb8, per_sample, rays, samples);
}

// Per-ray traversal work of a whole render (all bounces), for the bytes-per-ray model:
// counters[5] = nodes, tris, spheres, quads, rays.  Single-threaded.
void pto_render_counters(const pto_scene *s, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed, uint32_t pixel_stride, uint64_t *counters,
                         uint64_t *by_bounce /* [3][64]: rays, AABB tests, largest AABB-test count of one ray; may be null */) {
    const Camera &cam = s->sc->camera;
    Counters c;
    g_ctr = &c;
    size_t npix = (size_t)cam.image_width * cam.image_height;
    if (pixel_stride == 0) pixel_stride = 1;
    for (uint32_t stratum = stratum_begin; stratum < stratum_end; stratum++) {
        uint32_t s_u = stratum / cam.sqrt_spp, s_v = stratum % cam.sqrt_spp;
        for (size_t i = 0; i < npix; i += pixel_stride) {
            Rng rng = Rng::for_sample(seed, (uint32_t)i, stratum);
            Ray ray = cam.get_ray(rng, (float)(i % cam.image_width), (float)(i / cam.image_width), (float)s_u, (float)s_v);
            (void)s->sc->ray_color(rng, ray, cam.max_depth, nullptr);
        }
    }
    g_ctr = nullptr;
    counters[0] += c.nodes; counters[1] += c.tris; counters[2] += c.spheres; counters[3] += c.quads; counters[4] += c.rays;
    if (by_bounce)
        for (int k = 0; k < 64; k++) {
            by_bounce[k] = c.depth_rays[k];
            by_bounce[64 + k] = c.depth_nodes[k];
            by_bounce[128 + k] = c.depth_max_nodes[k];
        }
}

void pto_to_rgb8(const float *rgb, uint64_t npix, float divisor, uint8_t *out) {
    for (uint64_t i = 0; i < npix; i++) to_rgb(v3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]) / divisor, out + 3 * i);
}

// ---- known-answer hooks for single functions ------------------------------
int pto_refract(const float *w_o, const float *n, float eta, float *out) {
    V3 r;
    if (!refract(v3(w_o[0], w_o[1], w_o[2]), v3(n[0], n[1], n[2]), eta, r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}
float pto_fresnel_o(const float *w_o, const float *n, float eta) { return fresnel_o(v3(w_o[0], w_o[1], w_o[2]), v3(n[0], n[1], n[2]), eta); }
void pto_onb(const float *n, float *out9) {
    ONB o = ONB::make(v3(n[0], n[1], n[2]));
    out9[0] = o.m.c0.x; out9[1] = o.m.c0.y; out9[2] = o.m.c0.z;
    out9[3] = o.m.c1.x; out9[4] = o.m.c1.y; out9[5] = o.m.c1.z;
    out9[6] = o.m.c2.x; out9[7] = o.m.c2.y; out9[8] = o.m.c2.z;
}
// Material functions at a fixed shading point.  rec_in: [7] = u, v, p.xyz, front_face (0/1), reserved.
void pto_material_eval(const pto_scene *s, int32_t mat, const float *w_out, const float *w_in, const float *rec_in, float *bsdf3, float *pdf, float *emitted3) {
    const Material *m = s->sc->mat(mat);
    HitRecord rec;
    rec.u = rec_in[0]; rec.v = rec_in[1];
    rec.p = v3(rec_in[2], rec_in[3], rec_in[4]);
    rec.front_face = rec_in[5] != 0.0f;
    rec.normal = v3(0, 1, 0); rec.t = 1.0f; rec.material = mat;
    V3 wo = v3(w_out[0], w_out[1], w_out[2]), wi = v3(w_in[0], w_in[1], w_in[2]);
    Color b = s->sc->shading.bsdf_evaluate(*m, wo, wi, rec);
    bsdf3[0] = b.x; bsdf3[1] = b.y; bsdf3[2] = b.z;
    *pdf = s->sc->shading.scattering_pdf(*m, wo, wi, rec);
    Color e = s->sc->shading.emitted(*m, rec);
    emitted3[0] = e.x; emitted3[1] = e.y; emitted3[2] = e.z;
}
// One Material::scatter call with the generator keyed by (seed, pixel, stratum); returns draws consumed.
uint32_t pto_material_scatter(const pto_scene *s, int32_t mat, const float *w_out, const float *rec_in, uint64_t seed, uint32_t pixel, uint32_t stratum, float *w_in3) {
    const Material *m = s->sc->mat(mat);
    HitRecord rec;
    rec.u = rec_in[0]; rec.v = rec_in[1];
    rec.p = v3(rec_in[2], rec_in[3], rec_in[4]);
    rec.front_face = rec_in[5] != 0.0f;
    rec.normal = v3(0, 1, 0); rec.t = 1.0f; rec.material = mat;
    Rng rng = Rng::for_sample(seed, pixel, stratum);
    V3 w = s->sc->shading.scatter(*m, rng, v3(w_out[0], w_out[1], w_out[2]), rec);
    w_in3[0] = w.x; w_in3[1] = w.y; w_in3[2] = w.z;
    return rng.counter;
}
// The same per probe as pt_material_probe of include/ptcore.h (whose structs these are): scatter, then
// bsdf_evaluate / scattering_pdf at the scattered direction and at in.w_in, and emitted.
void pto_material_probe(const pto_scene *s, int32_t mat, uint64_t seed, const pt_material_probe_in *in, uint64_t n, pt_material_probe_out *out) {
    const Material *m = s->sc->mat(mat);
    const Shading &sh = s->sc->shading;
    for (uint64_t i = 0; i < n; i++) {
        const pt_material_probe_in &pi = in[i];
        HitRecord rec;
        rec.u = pi.u; rec.v = pi.v;
        rec.p = v3(pi.p[0], pi.p[1], pi.p[2]);
        rec.front_face = pi.front_face != 0;
        rec.normal = v3(0, 1, 0); rec.t = 1.0f; rec.material = mat;
        const V3 wo = v3(pi.w_out[0], pi.w_out[1], pi.w_out[2]), wa = v3(pi.w_in[0], pi.w_in[1], pi.w_in[2]);
        Rng rng = Rng::for_sample(seed, pi.pixel, pi.stratum);
        V3 wi = sh.scatter(*m, rng, wo, rec);
        pt_material_probe_out &o = out[i];
        o.w_in[0] = wi.x; o.w_in[1] = wi.y; o.w_in[2] = wi.z;
        o.draws = rng.counter;
        Color f = sh.bsdf_evaluate(*m, wo, wi, rec), fa = sh.bsdf_evaluate(*m, wo, wa, rec), e = sh.emitted(*m, rec);
        o.bsdf[0] = f.x; o.bsdf[1] = f.y; o.bsdf[2] = f.z;
        o.pdf = sh.scattering_pdf(*m, wo, wi, rec);
        o.bsdf_at[0] = fa.x; o.bsdf_at[1] = fa.y; o.bsdf_at[2] = fa.z;
        o.pdf_at = sh.scattering_pdf(*m, wo, wa, rec);
        o.emitted[0] = e.x; o.emitted[1] = e.y; o.emitted[2] = e.z;
        o.reserved = 0;
    }
}
void pto_texture_value(const pto_scene *s, int32_t tex, float u, float v, float *out3) {
    Color c = s->sc->textures.value(tex, u, v, v3(0, 0, 0));
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
float pto_light_pdf(const pto_scene *s, const float *origin, const float *dir) {
    return s->sc->lights.pdf(Ray::make(v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2])));
}
// LightBuffer::pdf (src/objects.rs:259-278) for n rays, rays[i][8] = origin, direction, (unused) t_min, t_max;
// solid_angle_tris selects the variant the next-event estimator uses
void pto_light_pdfs(const pto_scene *s, const float *rays, uint64_t n, int solid_angle_tris, float *out) {
    for (uint64_t i = 0; i < n; i++) {
        const float *r = rays + 8 * i;
        const Ray ray = Ray::make(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]));
        out[i] = solid_angle_tris ? s->sc->lights.pdf_nee(ray) : s->sc->lights.pdf(ray);
    }
}
// raw generator output, for checking the device generator against this one
void pto_rng_draws(uint64_t seed, uint32_t pixel, uint32_t stratum, uint32_t n, uint64_t *out) {
    Rng r = Rng::for_sample(seed, pixel, stratum);
    for (uint32_t i = 0; i < n; i++) out[i] = r.next_u64();
}

}  // extern "C"
