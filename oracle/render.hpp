// TEST INFRASTRUCTURE — CPU oracle (see na.hpp).
//
// Camera, the recursive integrator exactly as src/camera.rs writes it, an
// iterative restatement of the same estimator (what a wavefront device renderer
// computes), the scene loader and the multithreaded pass loop.
// parity unpinned: the reference holds no golden image or per-sample vector.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>

#include "shading.hpp"

namespace oracle {

// src/camera.rs:34-144
struct Camera {
    uint32_t image_width, image_height;
    float defocus_angle;
    V3 defocus_disk_u, defocus_disk_v;
    V3 center, pixel00, pixel_du, pixel_dv;
    uint32_t sqrt_spp;
    float inv_sqrt_spp;
    uint32_t max_depth;
    Color background;

    static Camera make(const pts_camera &c) {
        Camera k;
        float aspect_ratio = c.aspect_ratio;
        uint32_t image_width = c.image_width;
        // `(image_width as f32 / aspect_ratio) as u32`, then max(1, ..)
        uint32_t ih = TextureTable::f2u32((float)image_width / aspect_ratio);
        uint32_t image_height = ih > 1u ? ih : 1u;
        float theta = c.vfov * (PI_F / 180.0f);  // f32::to_radians
        float h = std::tan(theta / 2.0f);
        V3 vup = v3(0, 1, 0);
        V3 look_from = v3(c.look_from[0], c.look_from[1], c.look_from[2]);
        V3 look_at = v3(c.look_at[0], c.look_at[1], c.look_at[2]);
        V3 center = look_from;
        float viewport_height = 2.0f * h * c.focal_length;
        float viewport_width = viewport_height * ((float)image_width / (float)image_height);
        V3 w = normalize(look_from - look_at);
        V3 u = normalize(cross(vup, w));
        V3 v = cross(w, u);
        V3 viewport_u = viewport_width * u;
        V3 viewport_v = viewport_height * -v;
        V3 pixel_du = viewport_u / (float)image_width;
        V3 pixel_dv = viewport_v / (float)image_height;
        V3 viewport_upper_left = center - viewport_u / 2.0f - viewport_v / 2.0f - c.focal_length * w;
        V3 pixel00 = viewport_upper_left + 0.5f * pixel_du + 0.5f * pixel_dv;
        float defocus_radius = c.focal_length * std::tan(0.5f * ((c.defocus_angle / 2.0f) * (PI_F / 180.0f)));
        k.defocus_disk_u = u * defocus_radius;
        k.defocus_disk_v = v * defocus_radius;
        k.sqrt_spp = TextureTable::f2u32(std::sqrt((float)c.spp));
        k.inv_sqrt_spp = 1.0f / (float)k.sqrt_spp;
        k.image_width = image_width;
        k.image_height = image_height;
        k.defocus_angle = c.defocus_angle;
        k.center = center;
        k.pixel00 = pixel00;
        k.pixel_du = pixel_du;
        k.pixel_dv = pixel_dv;
        k.max_depth = c.max_depth;
        k.background = v3(c.background[0], c.background[1], c.background[2]);
        return k;
    }

    // :238-260
    Ray get_ray(Rng &rng, float u, float v, float s_u, float s_v) const {
        float disk_u, disk_v;
        for (;;) {  // DiskSampler::unit().sample, src/types/sampler.rs:49-58
            float x = rng.range(-1.0f, 1.0f);
            float y = rng.range(-1.0f, 1.0f);
            if (x * x + y * y < 1.0f) { disk_u = x; disk_v = y; break; }
        }
        V3 ray_center = (defocus_angle <= 0.0f) ? center : center + disk_u * defocus_disk_u + disk_v * defocus_disk_v;
        float sx = rng.range(0.0f, 1.0f);  // SquareSampler((0.5,0.5),0.5), sampler.rs:22-27
        float sy = rng.range(0.0f, 1.0f);
        float offset_u = (s_u + sx) * inv_sqrt_spp - 0.5f;
        float offset_v = (s_v + sy) * inv_sqrt_spp - 0.5f;
        V3 pixel_center = pixel00 + (u + offset_u) * pixel_du + (v + offset_v) * pixel_dv;
        return Ray::make(ray_center, pixel_center - ray_center);
    }
    // Deterministic primary ray: jitter fixed at 0.5 within a 1x1 stratum grid, no lens offset.
    Ray fixed_ray(float u, float v) const {
        float offset_u = (0.0f + 0.5f) * 1.0f - 0.5f;
        float offset_v = (0.0f + 0.5f) * 1.0f - 0.5f;
        V3 pixel_center = pixel00 + (u + offset_u) * pixel_du + (v + offset_v) * pixel_dv;
        return Ray::make(center, pixel_center - center);
    }
};

struct Scene {
    PrimitiveBuffer objects;
    LightBuffer lights;
    std::vector<Material> materials;
    TextureTable textures;
    Shading shading;
    Camera camera;
    float bsdf_sample_probability = 1.0f;  // BSDF_SAMPLE_PROBABILITY, src/camera.rs:317
    std::string error;
    std::vector<pts_medium> media;  // not in the reference, see pts_medium
    std::vector<pts_moving_sphere> moving;  // not in the reference, see pts_moving_sphere

    // The time of a path: a hash of its generator's key, none of its draws.
    static float path_time(const Rng &rng) { return (float)(uint32_t)(mix64(rng.key ^ 0x6D6F74696F6E5F74ull) >> 40) * (1.0f / 16777216.0f); }
    // PrimitiveBuffer::hit, then the moving spheres where they are at the path's time.
    bool hit_scene(const Rng &rng, Ray &ray, HitRecord &rec) const {
        bool any = objects.hit(ray, rec);
        if (moving.empty()) return any;
        const float time = path_time(rng);
        for (const pts_moving_sphere &m : moving) {
            V3 c0 = v3(m.center0[0], m.center0[1], m.center0[2]), c1 = v3(m.center1[0], m.center1[1], m.center1[2]);
            Sphere s = Sphere::make(c0 + time * (c1 - c0), m.radius, m.material);
            HitRecord h;
            if (s.hit(ray, h)) {
                rec = h;
                any = true;
            }
        }
        return any;
    }

    // Does the ray scatter in a medium before `t_closest` (its closest surface hit, or infinity)?  Media are
    // tried in index order, each against the closest event so far; a medium the ray crosses costs one draw.
    bool medium_event(Rng &rng, const Ray &ray, float t_closest, float &t_event, int &which) const {
        bool any = false;
        float best = t_closest;
        for (size_t m = 0; m < media.size(); m++) {
            const pts_medium &md = media[m];
            V3 oc = v3(md.center[0], md.center[1], md.center[2]) - ray.origin;
            float a = norm_squared(ray.direction);
            float h = dot(ray.direction, oc);
            float c = norm_squared(oc) - md.radius * md.radius;
            float disc = h * h - a * c;
            if (!(disc >= 0.0f)) continue;
            float sq = std::sqrt(disc);
            float t1 = (h - sq) / a, t2 = (h + sq) / a;
            if (t1 < ray.t_min) t1 = ray.t_min;
            if (t2 > best) t2 = best;
            if (!(t1 < t2)) continue;
            float len = std::sqrt(a);
            float inside = (t2 - t1) * len;
            float u = rng.range(0.0f, 1.0f);
            float dist = (-1.0f / md.density) * std::log(u);
            if (!(dist <= inside)) continue;
            best = t1 + dist / len;
            which = (int)m;
            any = true;
        }
        t_event = best;
        return any;
    }
    // The share of light that gets from the ray's origin to t_end without scattering (next-event estimation:
    // shadow rays are not scattered, they are dimmed): exp(-sum of density x length inside each medium).
    float transmittance(const Ray &ray, float t_end) const {
        float tau = 0.0f;
        for (const pts_medium &md : media) {
            V3 oc = v3(md.center[0], md.center[1], md.center[2]) - ray.origin;
            float a = norm_squared(ray.direction);
            float h = dot(ray.direction, oc);
            float c = norm_squared(oc) - md.radius * md.radius;
            float disc = h * h - a * c;
            if (!(disc >= 0.0f)) continue;
            float sq = std::sqrt(disc);
            float t1 = (h - sq) / a, t2 = (h + sq) / a;
            if (t1 < ray.t_min) t1 = ray.t_min;
            if (t2 > t_end) t2 = t_end;
            if (!(t1 < t2)) continue;
            tau = tau + md.density * ((t2 - t1) * std::sqrt(a));
        }
        return std::exp(-tau);
    }
    // The ray that leaves a scattering event in medium `which` at p: uniform on the sphere, two draws.
    Ray medium_scatter(Rng &rng, int which, V3 p, Color &albedo) const {
        albedo = textures.value(media[(size_t)which].texture, 0.0f, 0.0f, p);
        float u1 = rng.range(0.0f, 1.0f), u2 = rng.range(0.0f, 1.0f);
        float z = 1.0f - 2.0f * u1;
        float r = std::sqrt(rmax(0.0f, 1.0f - z * z));
        float phi = 2.0f * PI_F * u2;
        return Ray::make(p, v3(r * std::cos(phi), r * std::sin(phi), z));
    }

    const Material *mat(int32_t id) const { return (id >= 0 && (size_t)id < materials.size()) ? &materials[(size_t)id] : nullptr; }

    // ---- src/camera.rs:262-373, recursion kept
    Color ray_color(Rng &rng, Ray &ray_out, uint32_t max_depth, uint64_t *rays) const {
        const float C = 0.0f;
        if (max_depth == 0) return gray(0.0f);
        if (rays) (*rays)++;
        if (g_ctr) g_ctr->rays++;
        HitRecord rec;
        const uint64_t nodes_before = g_ctr ? g_ctr->nodes : 0;
        const bool any_hit = hit_scene(rng, ray_out, rec);
        if (g_ctr) {
            const uint32_t bounce = camera.max_depth >= max_depth ? camera.max_depth - max_depth : 0;
            const uint64_t n = g_ctr->nodes - nodes_before;
            if (bounce < 64) {
                g_ctr->depth_rays[bounce]++;
                g_ctr->depth_nodes[bounce] += n;
                if (n > g_ctr->depth_max_nodes[bounce]) g_ctr->depth_max_nodes[bounce] = n;
            }
        }
        if (!media.empty()) {
            float t_event;
            int which = 0;
            if (medium_event(rng, ray_out, any_hit ? rec.t : INFINITY, t_event, which)) {
                Color albedo;
                Ray ray_in = medium_scatter(rng, which, ray_out.at(t_event), albedo);
                return cmul(albedo, ray_color(rng, ray_in, max_depth - 1, rays));
            }
        }
        if (!any_hit) return camera.background;
        const Material *material = mat(rec.material);
        if (!material) return gray(0.0f);
        Color l_emitted = shading.emitted(*material, rec);
        if (material->is_emissive()) return l_emitted;
        ONB surface = ONB::make(rec.normal);
        V3 w_out = -normalize(surface.to_local(ray_out.direction));
        if (material->is_specular()) {
            V3 w_in = shading.scatter(*material, rng, w_out, rec);
            Color attenuation = shading.bsdf_evaluate(*material, w_out, w_in, rec);
            Ray ray_in = Ray::make(rec.p, surface.to_world(w_in));
            Color l_reflected = ray_color(rng, ray_in, max_depth - 1, rays);
            return l_emitted + cmul(attenuation, l_reflected);
        }
        const float P = bsdf_sample_probability;
        V3 w_in, w_in_world;
        bool is_bsdf;
        if (rng.gen_bool((double)P)) {
            w_in = shading.scatter(*material, rng, w_out, rec);
            is_bsdf = true;
            w_in_world = surface.to_world(w_in);
        } else {
            V3 light_sample = lights.sample(rng, rec.p);
            w_in = normalize(surface.to_local(light_sample));
            is_bsdf = false;
            w_in_world = light_sample;
        }
        uint32_t depth = is_bsdf ? max_depth : 2u;
        float scatter_pdf = shading.scattering_pdf(*material, w_out, w_in, rec);
        float light_pdf = lights.pdf(Ray::make(rec.p, w_in_world));
        float pdf = P * scatter_pdf + (1.0f - P) * light_pdf;
        Color bsdf = shading.bsdf_evaluate(*material, w_out, w_in, rec);
        float cosv = std::fabs(w_in.y);
        Color throughput = (bsdf * cosv) / pdf;
        float weight = !is_bsdf ? 0.0f : 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
        if (weight != weight || rng.gen_bool((double)weight)) return gray(C);
        Ray ray_in = Ray::make(rec.p, w_in_world);
        Color l_incoming = ray_color(rng, ray_in, depth - 1, rays);
        Color l_outgoing = l_emitted + cmul(throughput, l_incoming);
        return (l_outgoing - (weight * gray(C))) / (1.0f - weight);
    }

    // ---- the same estimator unrolled: one multiplicative factor per vertex, the
    // path contributes once at its end.  beta_k = throughput_k / (1 - weight_k) at
    // non-specular vertices and attenuation_k at specular ones.
    Color ray_color_iterative(Rng &rng, Ray ray, uint32_t max_depth, uint64_t *rays) const {
        Color beta = gray(1.0f);
        uint32_t depth = max_depth;
        for (;;) {
            if (depth == 0) return cmul(beta, gray(0.0f));
            if (rays) (*rays)++;
            HitRecord rec;
            const bool any_hit = hit_scene(rng, ray, rec);
            if (!media.empty()) {
                float t_event;
                int which = 0;
                if (medium_event(rng, ray, any_hit ? rec.t : INFINITY, t_event, which)) {
                    Color albedo;
                    Ray next = medium_scatter(rng, which, ray.at(t_event), albedo);
                    beta = cmul(beta, albedo);
                    ray = next;
                    depth -= 1;
                    continue;
                }
            }
            if (!any_hit) return cmul(beta, camera.background);
            const Material *material = mat(rec.material);
            if (!material) return cmul(beta, gray(0.0f));
            if (material->is_emissive()) return cmul(beta, shading.emitted(*material, rec));
            ONB surface = ONB::make(rec.normal);
            V3 w_out = -normalize(surface.to_local(ray.direction));
            if (material->is_specular()) {
                V3 w_in = shading.scatter(*material, rng, w_out, rec);
                beta = cmul(beta, shading.bsdf_evaluate(*material, w_out, w_in, rec));
                ray = Ray::make(rec.p, surface.to_world(w_in));
                depth -= 1;
                continue;
            }
            const float P = bsdf_sample_probability;
            V3 w_in, w_in_world;
            bool is_bsdf;
            if (rng.gen_bool((double)P)) {
                w_in = shading.scatter(*material, rng, w_out, rec);
                is_bsdf = true;
                w_in_world = surface.to_world(w_in);
            } else {
                V3 light_sample = lights.sample(rng, rec.p);
                w_in = normalize(surface.to_local(light_sample));
                is_bsdf = false;
                w_in_world = light_sample;
            }
            if (!is_bsdf) depth = 2u;
            float scatter_pdf = shading.scattering_pdf(*material, w_out, w_in, rec);
            float light_pdf = lights.pdf(Ray::make(rec.p, w_in_world));
            float pdf = P * scatter_pdf + (1.0f - P) * light_pdf;
            Color bsdf = shading.bsdf_evaluate(*material, w_out, w_in, rec);
            float cosv = std::fabs(w_in.y);
            Color throughput = (bsdf * cosv) / pdf;
            float weight = !is_bsdf ? 0.0f : 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
            if (weight != weight || rng.gen_bool((double)weight)) return gray(0.0f);
            beta = cmul(beta, throughput / (1.0f - weight));
            ray = Ray::make(rec.p, w_in_world);
            depth -= 1;
        }
    }

    // ---- not in the reference: next-event estimation with one LightBuffer::sample direction and the
    // balance heuristic at the vertices whose material has a scattering_pdf that is the density of its
    // scatter (Lambertian, DisneyDiffuse, DisneySheen; tests/test_material_pdfs.py); the other non-specular
    // materials are handled as ray_color handles them with BSDF_SAMPLE_PROBABILITY = 1.  This is the CPU
    // statement, operation for operation, of the estimator PT_OPT_INTEGRATOR = 1 of the device library
    // runs.  Radiance is summed along the path in the order it is found.  prev_pdf: density with which
    // the current ray was drawn at a vertex with light sampling, 0 otherwise.
    // Light sampling picks one light and a point on it; the sample counts with that light's own density over
    // directions (divided by the number of lights) and the light's emitted radiance where the ray meets it, if
    // nothing lies between the vertex and that point: the shadow ray ends just short of the light (SHADOW_REACH x the
    // distance) and any hit along it blocks the sample.  An emitter that a BSDF-sampled ray finds is weighted
    // against the density with which light sampling yields that very point (pdf_nee_at).  Participating media
    // scatter the path's own rays as in ray_color (no light is sampled at such an event) and dim the shadow rays
    // by their transmittance; moving spheres are where they are at the path's time for both kinds of ray.
    static constexpr float SHADOW_REACH = 0.999f;
    Color ray_color_nee(Rng &rng, Ray ray, uint32_t max_depth, uint64_t *rays) const {
        Color L = gray(0.0f), beta = gray(1.0f);
        float prev_pdf = 0.0f;
        uint32_t depth = max_depth;
        for (;;) {
            if (depth == 0) return L;
            if (rays) (*rays)++;
            HitRecord rec;
            const bool any_hit = hit_scene(rng, ray, rec);
            if (!media.empty()) {
                float t_event;
                int which = 0;
                if (medium_event(rng, ray, any_hit ? rec.t : INFINITY, t_event, which)) {
                    Color albedo;
                    Ray next = medium_scatter(rng, which, ray.at(t_event), albedo);
                    beta = cmul(beta, albedo);
                    ray = next;
                    prev_pdf = 0.0f;
                    depth -= 1;
                    continue;
                }
            }
            if (!any_hit) return L + cmul(beta, camera.background);
            const Material *material = mat(rec.material);
            if (!material) return L;
            if (material->is_emissive()) {
                float w = 1.0f;
                // the density with which light sampling draws this direction, whatever lies along it: a fresh ray, since
                // `ray` now ends at the emitter and a light behind it would no longer count
                if (lights.objects.len() != 0 && prev_pdf > 0.0f && prev_pdf < INFINITY)
                    w = prev_pdf / (prev_pdf + lights.pdf_nee_at(Ray::make(ray.origin, ray.direction), rec.t));
                return L + cmul(beta, shading.emitted(*material, rec)) * w;
            }
            ONB surface = ONB::make(rec.normal);
            V3 w_out = -normalize(surface.to_local(ray.direction));
            if (material->is_specular()) {
                V3 w_in = shading.scatter(*material, rng, w_out, rec);
                beta = cmul(beta, shading.bsdf_evaluate(*material, w_out, w_in, rec));
                ray = Ray::make(rec.p, surface.to_world(w_in));
                prev_pdf = 0.0f;
                depth -= 1;
                continue;
            }
            const bool light_sampling = material->type == PTS_MAT_LAMBERTIAN || material->type == PTS_MAT_DISNEY_DIFFUSE ||
                                        material->type == PTS_MAT_DISNEY_SHEEN;
            if (!light_sampling) {  // ray_color's vertex with P = 1
                (void)rng.gen_bool(1.0);
                V3 w_in = shading.scatter(*material, rng, w_out, rec);
                V3 w_in_world = surface.to_world(w_in);
                float scatter_pdf = shading.scattering_pdf(*material, w_out, w_in, rec);
                float light_pdf = lights.pdf(Ray::make(rec.p, w_in_world));
                float pdf = 1.0f * scatter_pdf + (1.0f - 1.0f) * light_pdf;
                Color bsdf = shading.bsdf_evaluate(*material, w_out, w_in, rec);
                float cosv = std::fabs(w_in.y);
                Color throughput = (bsdf * cosv) / pdf;
                float weight = 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
                if (weight != weight || rng.gen_bool((double)weight)) return L;
                beta = cmul(beta, throughput / (1.0f - weight));
                prev_pdf = 0.0f;
                ray = Ray::make(rec.p, w_in_world);
                depth -= 1;
                continue;
            }
            if (lights.objects.len() != 0) {
                size_t picked = 0;
                V3 ls = lights.sample_nee(rng, rec.p, &picked);
                Ray shadow = Ray::make(rec.p, ls);
                HitRecord lrec;
                float lp = lights.term_nee(picked, shadow, lrec) / (float)lights.objects.len();
                if (lp > 0.0f && lp < INFINITY) {
                    V3 wl = normalize(surface.to_local(ls));
                    float sp = shading.scattering_pdf(*material, w_out, wl, rec);
                    if (sp > 0.0f) {
                        Color f = shading.bsdf_evaluate(*material, w_out, wl, rec) * std::fabs(wl.y);
                        Color c = cmul(beta, f) / (lp + sp);
                        const Material *lm = mat(lrec.material);
                        Color le = (lm && lm->is_emissive()) ? shading.emitted(*lm, lrec) : gray(0.0f);
                        if (le.x != 0.0f || le.y != 0.0f || le.z != 0.0f) {
                            if (rays) (*rays)++;
                            shadow.t_max = lrec.t * SHADOW_REACH;
                            HitRecord srec;
                            if (!hit_scene(rng, shadow, srec)) {
                                Color add = cmul(c, le);
                                if (!media.empty()) add = add * transmittance(shadow, shadow.t_max);
                                L = L + add;
                            }
                        }
                    }
                }
            }
            V3 w_in = shading.scatter(*material, rng, w_out, rec);
            V3 w_in_world = surface.to_world(w_in);
            float sp = shading.scattering_pdf(*material, w_out, w_in, rec);
            Color bsdf = shading.bsdf_evaluate(*material, w_out, w_in, rec);
            float cosv = std::fabs(w_in.y);
            Color throughput = (bsdf * cosv) / sp;
            float weight = 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
            if (weight != weight || rng.gen_bool((double)weight)) return L;
            beta = cmul(beta, throughput / (1.0f - weight));
            prev_pdf = sp;
            ray = Ray::make(rec.p, w_in_world);
            depth -= 1;
        }
    }
};

// src/types/color.rs:36-42, :68-78
inline float linear_to_gamma(float c) {
    if (c < 0.0031308f) return 12.92f * c;
    return 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}
inline void to_rgb(Color c, uint8_t out[3]) {
    uint32_t ir = TextureTable::f2u32(255.999f * linear_to_gamma(c.x));
    uint32_t ig = TextureTable::f2u32(255.999f * linear_to_gamma(c.y));
    uint32_t ib = TextureTable::f2u32(255.999f * linear_to_gamma(c.z));
    out[0] = (uint8_t)(ir > 255 ? 255 : ir);
    out[1] = (uint8_t)(ig > 255 ? 255 : ig);
    out[2] = (uint8_t)(ib > 255 ? 255 : ib);
}

// ---- scene loader: authoring-level description -> reference data structures
inline bool load_primset(const pts_scene &s, uint32_t set_index, PrimitiveBuffer &out, std::string &err, int depth = 0);

inline Instance load_instance(const pts_scene &s, const pts_instance &pi, std::string &err, int depth) {
    auto child = std::make_shared<PrimitiveBuffer>();
    if (pi.child_set >= s.n_sets) { err = "instance child_set out of range"; return Instance::from_obj(child); }
    load_primset(s, pi.child_set, *child, err, depth + 1);
    Instance inst = Instance::from_obj(child);
    for (uint32_t k = 0; k < pi.op_count; k++) {
        const pts_xform_op &op = s.ops[pi.op_first + k];
        const float *a = op.a;
        switch (op.op) {
            case PTS_OP_ROTATE: inst.rotate(a[0], v3(a[1], a[2], a[3])); break;
            case PTS_OP_ROTATE_X: inst.rotate_x(a[0]); break;
            case PTS_OP_ROTATE_Y: inst.rotate_y(a[0]); break;
            case PTS_OP_ROTATE_Z: inst.rotate_z(a[0]); break;
            case PTS_OP_TRANSLATE: inst.translate(v3(a[0], a[1], a[2])); break;
            case PTS_OP_TRANSLATE_X: inst.translate(v3(a[0], 0, 0)); break;
            case PTS_OP_TRANSLATE_Y: inst.translate(v3(0, a[0], 0)); break;
            case PTS_OP_TRANSLATE_Z: inst.translate(v3(0, 0, a[0])); break;
            case PTS_OP_SCALE: inst.scale(v3(a[0], a[1], a[2])); break;
            // scale_x(x): inverse uses (1/x, 1, 1) (src/objects.rs:356-361)
            case PTS_OP_SCALE_X: inst.transform = mul(m4_scaling(v3(a[0], 1, 1)), inst.transform); inst.inverse = mul(inst.inverse, m4_scaling(v3(1.0f / a[0], 1, 1))); break;
            case PTS_OP_SCALE_Y: inst.transform = mul(m4_scaling(v3(1, a[0], 1)), inst.transform); inst.inverse = mul(inst.inverse, m4_scaling(v3(1, 1.0f / a[0], 1))); break;
            case PTS_OP_SCALE_Z: inst.transform = mul(m4_scaling(v3(1, 1, a[0])), inst.transform); inst.inverse = mul(inst.inverse, m4_scaling(v3(1, 1, 1.0f / a[0]))); break;
            case PTS_OP_SCALE_UNIFORM: {
                float sc = a[0], isc = 1.0f / a[0];
                inst.transform = mul(m4_scaling(v3(sc, sc, sc)), inst.transform);
                inst.inverse = mul(inst.inverse, m4_scaling(v3(isc, isc, isc)));
                break;
            }
            default: err = "unknown transform op"; break;
        }
    }
    return inst;
}

inline bool load_primset(const pts_scene &s, uint32_t set_index, PrimitiveBuffer &out, std::string &err, int depth) {
    if (depth > 8) { err = "instance nesting too deep"; return false; }
    const pts_primset &ps = s.sets[set_index];
    for (uint32_t i = ps.tris.first; i < ps.tris.first + ps.tris.count; i++) {
        V3 v[3], n[3];
        float uv[3][2];
        for (int c = 0; c < 3; c++) v[c] = v3(s.tri_pos[9 * i + 3 * c], s.tri_pos[9 * i + 3 * c + 1], s.tri_pos[9 * i + 3 * c + 2]);
        uint8_t flags = s.tri_flags ? s.tri_flags[i] : (uint8_t)((s.tri_nrm ? PTS_TRI_HAS_NORMALS : 0) | (s.tri_uv ? PTS_TRI_HAS_UVS : 0));
        bool hn = (flags & PTS_TRI_HAS_NORMALS) && s.tri_nrm, hu = (flags & PTS_TRI_HAS_UVS) && s.tri_uv;
        if (hn) for (int c = 0; c < 3; c++) n[c] = v3(s.tri_nrm[9 * i + 3 * c], s.tri_nrm[9 * i + 3 * c + 1], s.tri_nrm[9 * i + 3 * c + 2]);
        if (hu) for (int c = 0; c < 3; c++) { uv[c][0] = s.tri_uv[6 * i + 2 * c]; uv[c][1] = s.tri_uv[6 * i + 2 * c + 1]; }
        out.add_triangle(Triangle::make(v, hn ? n : nullptr, hu ? uv : nullptr, s.tri_mat[i]));
    }
    for (uint32_t i = ps.spheres.first; i < ps.spheres.first + ps.spheres.count; i++)
        out.add_sphere(Sphere::make(v3(s.spheres[4 * i], s.spheres[4 * i + 1], s.spheres[4 * i + 2]), s.spheres[4 * i + 3], s.sphere_mat[i]));
    for (uint32_t i = ps.quads.first; i < ps.quads.first + ps.quads.count; i++) {
        const float *q = &s.quads[9 * i];
        out.add_quad(Quad::make(v3(q[0], q[1], q[2]), v3(q[3], q[4], q[5]), v3(q[6], q[7], q[8]), s.quad_mat[i]));
    }
    for (uint32_t i = ps.instances.first; i < ps.instances.first + ps.instances.count; i++)
        out.add_instance(load_instance(s, s.instances[i], err, depth));
    if (ps.build_bvh && !out.build_bvh()) { err = "BVH deeper than 63 levels"; return false; }
    return err.empty();
}

inline Scene *load_scene(const pts_scene &s, const pts_camera &cam, std::string &err) {
    auto *sc = new Scene();
    sc->textures.tex.assign(s.textures, s.textures + s.n_textures);
    sc->textures.images.assign(s.images, s.images + s.n_images);
    sc->textures.texels.assign(s.texels, s.texels + s.n_texel_bytes);
    for (uint32_t i = 0; i < s.n_materials; i++) sc->materials.push_back(Material::make(s.materials[i]));
    sc->shading.tt = &sc->textures;
    sc->camera = Camera::make(cam);
    if (s.n_sets < 2) { err = "need at least the object set and the light set"; delete sc; return nullptr; }
    load_primset(s, 0, sc->objects, err);
    load_primset(s, 1, sc->lights.objects, err);
    if (s.n_media > PTS_MAX_MEDIA) err = "too many participating media";
    else sc->media.assign(s.media, s.media + s.n_media);
    if (s.n_moving > PTS_MAX_MOVING) err = "too many moving spheres";
    else sc->moving.assign(s.moving, s.moving + s.n_moving);
    if (!err.empty()) { delete sc; return nullptr; }
    return sc;
}

// src/camera.rs:162-236: serial stratified passes, pixels in parallel.
struct RenderOptions {
    uint32_t stratum_begin, stratum_end;
    uint64_t seed;
    uint32_t threads;
    int iterative;          // 0 = recursive form (the reference's), 1 = unrolled form, 2 = next-event estimation (ray_color_nee)
    int tonemap_each_pass;  // reference refreshes the whole 8-bit image after every pass
};


// A persistent worker pool (the reference runs on rayon's global pool, which is
// also created once): run(f) executes f on every worker and on the caller.
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void()> job;
    uint64_t gen = 0;
    unsigned pending = 0;
    bool stop = false;
    explicit Pool(unsigned n) {
        for (unsigned t = 1; t < n; t++)
            th.emplace_back([this] {
                uint64_t seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> l(m);
                    cv.wait(l, [&] { return stop || gen != seen; });
                    if (stop) return;
                    seen = gen;
                    auto j = job;
                    l.unlock();
                    j();
                    l.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void run(const std::function<void()> &f) {
        { std::lock_guard<std::mutex> l(m); job = f; pending = (unsigned)th.size(); gen++; }
        cv.notify_all();
        f();
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return pending == 0; });
    }
};

inline void render(const Scene &sc, const RenderOptions &o, float *accum /*W*H*3, added to*/, uint8_t *rgb8 /*may be null*/,
                   float *per_sample /*(n_strata, W*H, 3) or null*/, uint64_t *rays_out, uint32_t *samples_inout) {
    const Camera &cam = sc.camera;
    const size_t npix = (size_t)cam.image_width * cam.image_height;
    unsigned nthreads = o.threads ? o.threads : std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    std::atomic<uint64_t> rays_total{0};
    Pool pool(nthreads);
    uint32_t samples = samples_inout ? *samples_inout : 0;
    for (uint32_t stratum = o.stratum_begin; stratum < o.stratum_end; stratum++) {
        // sample_list is s_u-major (:176-181)
        uint32_t s_u = stratum / cam.sqrt_spp, s_v = stratum % cam.sqrt_spp;
        samples += 1;
        std::atomic<size_t> next{0};
        const size_t chunk = 1024;
        auto worker = [&]() {
            uint64_t rays = 0;
            for (;;) {
                size_t b = next.fetch_add(chunk);
                if (b >= npix) break;
                size_t e = std::min(npix, b + chunk);
                for (size_t i = b; i < e; i++) {
                    Rng rng = Rng::for_sample(o.seed, (uint32_t)i, stratum);
                    size_t u = i % cam.image_width, v = i / cam.image_width;
                    Ray ray = cam.get_ray(rng, (float)u, (float)v, (float)s_u, (float)s_v);
                    Color c = o.iterative == 2 ? sc.ray_color_nee(rng, ray, cam.max_depth, &rays)
                              : o.iterative ? sc.ray_color_iterative(rng, ray, cam.max_depth, &rays)
                                            : sc.ray_color(rng, ray, cam.max_depth, &rays);
                    if (per_sample) {
                        float *ps = per_sample + ((size_t)(stratum - o.stratum_begin) * npix + i) * 3;
                        ps[0] = c.x; ps[1] = c.y; ps[2] = c.z;
                    }
                    if (c.x != c.x || c.y != c.y || c.z != c.z) continue;  // :212-217
                    accum[3 * i] += c.x;
                    accum[3 * i + 1] += c.y;
                    accum[3 * i + 2] += c.z;
                }
            }
            rays_total += rays;
        };
        pool.run(worker);
        if (rgb8 && o.tonemap_each_pass) {  // :221-226
            std::atomic<size_t> nx{0};
            auto tm = [&]() {
                for (;;) {
                    size_t b = nx.fetch_add(4096);
                    if (b >= npix) break;
                    size_t e = std::min(npix, b + 4096);
                    for (size_t i = b; i < e; i++) {
                        Color c = v3(accum[3 * i], accum[3 * i + 1], accum[3 * i + 2]) / (float)samples;
                        to_rgb(c, rgb8 + 3 * i);
                    }
                }
            };
            pool.run(tm);
        }
    }
    if (rgb8 && !o.tonemap_each_pass && samples > 0) {
        for (size_t i = 0; i < npix; i++) {
            Color c = v3(accum[3 * i], accum[3 * i + 1], accum[3 * i + 2]) / (float)samples;
            to_rgb(c, rgb8 + 3 * i);
        }
    }
    if (rays_out) *rays_out += rays_total.load();
    if (samples_inout) *samples_inout = samples;
}

}  // namespace oracle
