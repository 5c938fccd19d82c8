// TEST INFRASTRUCTURE — CPU oracle (see na.hpp).
//
// Textures and materials of the reference, restated with their quirks intact
// (the Disney lobes in particular are reproduced as written, not corrected).
// Local shading frame is y-up; w_out.y >= 0 by construction.
// parity unpinned: src/materials.rs:42-66 holds the reference's only two tests and
// neither asserts anything.
#pragma once
#include <vector>

#include "../include/ptscene_src.h"
#include "geom.hpp"

namespace oracle {

typedef V3 Color;
inline Color gray(float v) { return v3(v, v, v); }
constexpr float INV_PI = 1.0f / PI_F;  // disney.rs:24

// ---------------------------------------------------------------- textures
struct TextureTable {
    std::vector<pts_texture> tex;
    std::vector<pts_image> images;
    std::vector<uint8_t> texels;

    // src/textures.rs:59-103, src/textures/image.rs:31-65
    // PTS_TEX_NOISE (not in the reference): gradient noise, 7 octaves; see include/ptscene_src.h
    static int32_t f2i32(float f) {
        if (f != f) return 0;
        if (f >= 2147483648.0f) return INT32_MAX;
        if (f <= -2147483648.0f) return INT32_MIN;
        return (int32_t)f;
    }
    static V3 lattice_gradient(uint32_t seed, int32_t ix, int32_t iy, int32_t iz) {
        uint64_t z = (uint64_t)seed ^ ((uint64_t)(uint32_t)ix * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)iy * 0xC2B2AE3D27D4EB4Full) ^
                     ((uint64_t)(uint32_t)iz * 0x165667B19E3779F9ull);
        z = mix64(z);
        V3 g = v3((float)(uint32_t)(z & 0x1FFFFFull) / 1048576.0f - 1.0f, (float)(uint32_t)((z >> 21) & 0x1FFFFFull) / 1048576.0f - 1.0f,
                  (float)(uint32_t)((z >> 42) & 0x1FFFFFull) / 1048576.0f - 1.0f);
        float n2 = dot(g, g);
        if (!(n2 > 0.0f)) return v3(1.0f, 0.0f, 0.0f);
        return g / std::sqrt(n2);
    }
    static float noise(uint32_t seed, V3 p) {
        float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
        int32_t ix = f2i32(fx), iy = f2i32(fy), iz = f2i32(fz);
        float u = p.x - fx, v = p.y - fy, w = p.z - fz;
        float su = u * u * (3.0f - 2.0f * u), sv = v * v * (3.0f - 2.0f * v), sw = w * w * (3.0f - 2.0f * w);
        float acc = 0.0f;
        for (uint32_t di = 0; di < 2; di++)
            for (uint32_t dj = 0; dj < 2; dj++)
                for (uint32_t dk = 0; dk < 2; dk++) {
                    V3 g = lattice_gradient(seed, (int32_t)((uint32_t)ix + di), (int32_t)((uint32_t)iy + dj), (int32_t)((uint32_t)iz + dk));
                    V3 d = v3(u - (float)di, v - (float)dj, w - (float)dk);
                    float weight = ((di ? su : 1.0f - su) * (dj ? sv : 1.0f - sv)) * (dk ? sw : 1.0f - sw);
                    acc += weight * dot(g, d);
                }
        return acc;
    }
    static float turbulence(uint32_t seed, V3 p) {
        float accum = 0.0f, weight = 1.0f;
        for (int i = 0; i < 7; i++) {
            accum += weight * noise(seed, p);
            weight *= 0.5f;
            p = p * 2.0f;
        }
        return std::fabs(accum);
    }

    Color value(int32_t id, float u, float v, V3 p) const {
        const pts_texture &t = tex[(size_t)id];
        switch (t.type) {
            case PTS_TEX_NOISE:
                return v3(t.color[0], t.color[1], t.color[2]) * (1.0f + std::sin(t.scale * p.z + 10.0f * turbulence((uint32_t)t.odd, p)));
            case PTS_TEX_SOLID:
                return v3(t.color[0], t.color[1], t.color[2]);
            case PTS_TEX_CHECKERED: {
                int64_t ui = f2i64(std::floor(u * t.scale));
                int64_t vi = f2i64(std::floor(v * t.scale));
                // wrapping add is irrelevant at these magnitudes; `%` keeps the sign
                if ((ui + vi) % 2 == 0) return value(t.even, u, v, p);
                return value(t.odd, u, v, p);
            }
            case PTS_TEX_IMAGE: {
                if (t.image < 0) {  // image.rs:54-63
                    int64_t ui = f2i64(std::floor(u * 64.0f));
                    int64_t vi = f2i64(std::floor(v * 64.0f));
                    return ((ui + vi) % 2 == 0) ? gray(1.0f) : gray(0.7f);
                }
                const pts_image &im = images[(size_t)t.image];
                if (im.height == 0 || im.width == 0) return gray(0.0f);
                float uu = rclamp(u, 0.0f, 1.0f);
                float vv = 1.0f - rclamp(v, 0.0f, 1.0f);
                uint32_t i = f2u32(uu * (float)(im.width - 1));
                uint32_t j = f2u32(vv * (float)(im.height - 1));
                const uint8_t *px = &texels[im.offset + 3ull * ((uint64_t)j * im.width + i)];
                return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
            }
        }
        return gray(0.0f);
    }
    // Rust `as i64` / `as u32` from f32: saturating, NaN -> 0
    static int64_t f2i64(float f) {
        if (f != f) return 0;
        if (f >= 9.2233720368547758e18f) return INT64_MAX;
        if (f <= -9.2233720368547758e18f) return INT64_MIN;
        return (int64_t)f;
    }
    static uint32_t f2u32(float f) {
        if (!(f > 0.0f)) return 0;
        if (f >= 4294967296.0f) return UINT32_MAX;
        return (uint32_t)f;
    }
};

// ------------------------------------------------- free helpers (materials.rs)
inline V3 reflect(V3 v, V3 n) {  // :16-19
    V3 m = -v;
    return m - 2.0f * dot(m, n) * n;
}
inline V3 reflect_y(V3 v) { return v3(v.x, -v.y, v.z); }  // :21-26
inline bool refract(V3 w_o, V3 n, float eta, V3 &out) {     // :28-40
    float cos_o = rmin(dot(w_o, n), 1.0f);
    float sin2_o = 1.0f - cos_o * cos_o;
    float sin2_i = sin2_o * eta * eta;
    if (sin2_i >= 1.0f) return false;
    float cos_i = std::sqrt(1.0f - sin2_i);
    out = normalize((eta * -w_o) + ((eta * cos_o) - cos_i) * n);
    return true;
}
inline V3 refract_y(V3 uv, float etai_over_etat) {  // :68-74
    float cos_theta = -rmin(uv.y, 1.0f);
    V3 r_out_perp = etai_over_etat * (uv + (cos_theta * v3(0, 1, 0)));
    V3 r_out_parallel = -std::sqrt(std::fabs(1.0f - norm_squared(r_out_perp))) * v3(0, 1, 0);
    return r_out_perp + r_out_parallel;
}

// ------------------------------------------------- disney.rs:26-74
inline float r_0(float ior) { return ((ior - 1.0f) * (ior - 1.0f)) / ((ior + 1.0f) * (ior + 1.0f)); }
inline float schlick_weight(float h_dot_l) { return std::pow(1.0f - h_dot_l, 5.0f); }
inline float schlick(float ior, float h_dot_l) {
    float r0 = r_0(ior);
    return r0 + (1.0f - r0) * schlick_weight(h_dot_l);
}
inline float fresnel_o(V3 w_o, V3 n, float eta) {  // :46-64
    float cos_o = dot(w_o, n);
    float sin2_o = rmax(1.0f - cos_o * cos_o, 0.0f);
    float sin2_i = sin2_o * eta * eta;
    if (sin2_i >= 1.0f) return 1.0f;
    cos_o = std::fabs(cos_o);
    float cos_i = std::sqrt(rmax(1.0f - sin2_i, 0.0f));
    eta = 1.0f / eta;
    float r_para = (cos_o - eta * cos_i) / (cos_o + eta * cos_i);
    float r_perp = (eta * cos_o - cos_i) / (eta * cos_o + eta * cos_i);
    return 0.5f * (r_para * r_para + r_perp * r_perp);
}
inline V3 calculate_tint(V3 base_color) {  // :67-74
    float luminance = rclamp(dot(v3(0.3f, 0.6f, 1.0f), base_color), 0.0f, 1.0f);
    if (luminance > 0.0f) return (1.0f / luminance) * base_color;
    return v3(1, 1, 1);
}

// cosine-weighted hemisphere sample shared by Lambertian, DisneyDiffuse, DisneySheen
inline V3 cosine_sample(Rng &rng) {  // lambertian.rs:47-57, disney.rs:370-380, :1051-1061
    float r1 = rng.range(0.0f, 1.0f);
    float r2 = rng.range(0.0f, 1.0f);
    float phi = 2.0f * PI_F * r1;
    float x = std::cos(phi) * std::sqrt(r2);
    float z = std::sin(phi) * std::sqrt(r2);
    float y = std::sqrt(1.0f - r2);
    return v3(x, y, z);
}

// src/types/pdf.rs:24-36 (used by Metal fuzz)
inline V3 sphere_pdf_generate(Rng &rng) {
    for (;;) {
        float u = rng.range(0.0f, 1.0f), v = rng.range(0.0f, 1.0f);
        float ns = u * u + v * v;
        if (ns <= 1.0f) {
            float x = 2.0f * u * std::sqrt(1.0f - ns);
            float y = 2.0f * v * std::sqrt(1.0f - ns);
            float z = 1.0f - (2.0f * ns);
            return v3(x, y, z);
        }
    }
}

// ------------------------------------------------- Disney sub-lobes
struct MicrofacetAlpha {  // disney.rs:537-553 / :721-737
    float a_x, a_z, inv_a_x, inv_a_z;
    static MicrofacetAlpha make(float roughness, float anisotropic) {
        const float A_MIN = 0.0001f;
        float aspect = std::sqrt(1.0f - 0.9f * anisotropic);
        float inv_aspect = 1.0f / aspect;
        MicrofacetAlpha m;
        m.a_x = rmax(roughness * roughness * inv_aspect, A_MIN);
        m.a_z = rmax(roughness * roughness * aspect, A_MIN);
        m.inv_a_x = 1.0f / m.a_x;
        m.inv_a_z = 1.0f / m.a_z;
        return m;
    }
    float ggx_metal(V3 h) const {  // :575-580
        float k = (h.x * h.x * inv_a_x * inv_a_x) + (h.z * h.z * inv_a_z * inv_a_z) + h.y * h.y;
        return INV_PI * inv_a_x * inv_a_z * (1.0f / (k * k));
    }
    float ggx_glass(V3 h) const {  // :765-769
        V3 h2 = cmul(h, h);
        float k = (h2.x * inv_a_x * inv_a_x) + (h2.z * inv_a_z * inv_a_z) + h2.y;
        return INV_PI * inv_a_x * inv_a_z * (1.0f / (k * k));
    }
    float separable_g(V3 w) const {  // :582-588, :771-777
        float x = a_x * w.x, y = w.y, z = a_z * w.z;
        return 2.0f / (1.0f + std::sqrt(1.0f + ((x * x) + (z * z)) / (y * y)));
    }
    // Heitz VNDF micro-normal, :660-688 and :866-894 (identical text)
    V3 sample_ne(Rng &rng, V3 w_out) const {
        float u1 = rng.range(0.0f, 1.0f);
        float u2 = rng.range(0.0f, 1.0f);
        V3 w_hemi = normalize(v3(a_x * w_out.x, w_out.y, a_z * w_out.z));
        V3 basis_1 = (w_hemi.y < 0.9999f) ? normalize(cross(v3(0, 1, 0), w_hemi)) : v3(1, 0, 0);
        V3 basis_2 = cross(w_hemi, basis_1);
        float r = std::sqrt(u1);
        float phi = 2.0f * PI_F * u2;
        float t1 = r * std::cos(phi);
        float t2 = r * std::sin(phi);
        float s = 0.5f * (1.0f + w_hemi.y);
        t2 = (1.0f - s) * std::sqrt(1.0f - t1 * t1) + (s * t2);
        V3 n_h = (t1 * basis_1) + (t2 * basis_2) + std::sqrt(rmax(1.0f - (t1 * t1) - (t2 * t2), 0.0f)) * w_hemi;
        return normalize(v3(a_x * n_h.x, rmax(n_h.y, 0.0f), a_z * n_h.z));
    }
};

struct Material {
    uint32_t type;
    int32_t texture;
    float p[14];
    // derived
    MicrofacetAlpha mf;       // DisneyMetal / DisneyGlass / Disney
    float cc_a2, cc_a2_invlog2;  // DisneyClearcoat (:403-413)

    static Material make(const pts_material &m) {
        Material r;
        r.type = m.type;
        r.texture = m.texture;
        for (int i = 0; i < 14; i++) r.p[i] = m.p[i];
        r.mf = MicrofacetAlpha{0, 0, 0, 0};
        r.cc_a2 = r.cc_a2_invlog2 = 0.0f;
        switch (m.type) {
            case PTS_MAT_METAL: r.p[3] = rclamp(m.p[3], 0.0f, 1.0f); break;           // metal.rs:17
            case PTS_MAT_DISNEY:
                r.p[2] = rmax(m.p[2], 0.0001f);                                        // disney.rs:123
                r.mf = MicrofacetAlpha::make(m.p[5], m.p[6]);
                r.set_clearcoat(m.p[9], m.p[10]);
                break;
            case PTS_MAT_DISNEY_METAL:
            case PTS_MAT_DISNEY_GLASS: r.mf = MicrofacetAlpha::make(m.p[0], m.p[1]); break;
            case PTS_MAT_DISNEY_CLEARCOAT: r.set_clearcoat(m.p[0], m.p[1]); break;
            default: break;
        }
        return r;
    }
    void set_clearcoat(float clearcoat, float gloss) {
        (void)clearcoat;
        float a = (1.0f - gloss) * 0.1f + (gloss * 0.001f);
        cc_a2 = a * a;
        cc_a2_invlog2 = 1.0f / std::log2(cc_a2);
    }
    bool is_emissive() const { return type == PTS_MAT_DIFFUSE_LIGHT; }  // light.rs:34-36
    bool is_specular() const {                                           // metal.rs:27, dielectric.rs:30, light.rs:38
        return type == PTS_MAT_METAL || type == PTS_MAT_DIELECTRIC || type == PTS_MAT_DIFFUSE_LIGHT;
    }
};

struct Shading {
    const TextureTable *tt;

    Color tex(const Material &m, const HitRecord &rec) const { return tt->value(m.texture, rec.u, rec.v, rec.p); }

    // ---- DisneyDiffuse :989-1032
    Color evaluate_diffuse(const Material &m, float roughness, float subsurface, V3 w_o, V3 h, V3 w_i, const HitRecord &rec) const {
        float cos_o = rmin(std::fabs(w_o.y), 1.0f);
        float cos_i = rmin(std::fabs(w_i.y), 1.0f);
        float cos_d = rmin(std::fabs(dot(h, w_i)), 1.0f);
        float fresnel_i = schlick_weight(cos_i);
        float fresnel_ow = schlick_weight(cos_o);
        float f_base = 0.0f;
        if (subsurface < 1.0f) {
            float f_d_90 = 0.5f + (2.0f * cos_d * cos_d * roughness);
            f_base = (1.0f + (f_d_90 - 1.0f) * fresnel_i) * (1.0f + (f_d_90 - 1.0f) * fresnel_ow);
        }
        float f_subsurf = 0.0f;
        if (subsurface > 0.0f) {
            float f_ss_90 = cos_d * cos_d * roughness;
            float f_ss = (1.0f + (f_ss_90 - 1.0f) * fresnel_ow) * (1.0f + (f_ss_90 - 1.0f) * fresnel_i);
            f_subsurf = 1.25f * (f_ss * ((1.0f / (cos_i + cos_o)) - 0.5f) + 0.5f);
        }
        float f_diffuse = ((1.0f - subsurface) * f_base) + (subsurface * f_subsurf);
        Color albedo = tex(m, rec);
        return f_diffuse * albedo * INV_PI;
    }
    // ---- DisneySheen :331-351
    Color evaluate_sheen(const Material &m, float sheen, float sheen_tint, V3 h, V3 w_i, const HitRecord &rec) const {
        if (sheen <= 0.0f) return gray(0.0f);
        Color base_color = tex(m, rec);
        float h_dot_w_i = dot(h, w_i);
        V3 tint = calculate_tint(base_color);
        return sheen * lerp(v3(1, 1, 1), tint, sheen_tint) * schlick_weight(h_dot_w_i);
    }
    // ---- DisneyClearcoat :415-465
    static float gtr1(const Material &m, float cos_d, float a2) {
        if (a2 >= 1.0f) return INV_PI;
        return INV_PI * (a2 - 1.0f) * m.cc_a2_invlog2 / (1.0f + (a2 - 1.0f) * cos_d * cos_d);
    }
    static float clearcoat_g(V3 w) {
        float x = 0.25f * w.x, y = w.y, z = 0.25f * w.z;
        return 2.0f / std::sqrt(1.0f + (((x * x) + (z * z)) / (y * y)));
    }
    Color evaluate_clearcoat(const Material &m, float clearcoat, V3 w_o, V3 h, V3 w_i) const {
        if (clearcoat <= 0.0f) return gray(0.0f);
        float cos_o = rmin(std::fabs(w_o.y), 1.0f);
        float cos_i = rmin(std::fabs(w_i.y), 1.0f);
        float cos_d = rmin(std::fabs(dot(h, w_i)), 1.0f);
        float a2 = m.cc_a2;
        float d = gtr1(m, cos_d, a2);
        float f = schlick(1.5f, cos_d);
        float g = clearcoat_g(w_i) * clearcoat_g(w_o);
        float value = (0.25f * clearcoat * d * f * g) / (cos_o * cos_i);
        return gray(value);
    }
    // ---- DisneyMetal :590-631
    Color evaluate_metal_color(const Material &m, V3 w_o, V3 h, V3 w_i, Color base_color) const {
        float cos_o = rmin(std::fabs(w_o.y), 1.0f);
        float cos_i = rmin(std::fabs(w_i.y), 1.0f);
        float cos_d = rmin(std::fabs(dot(h, w_i)), 1.0f);
        Color f = lerp(base_color, v3(1, 1, 1), schlick_weight(cos_d));
        float d = m.mf.ggx_metal(h);
        float g = m.mf.separable_g(w_o) * m.mf.separable_g(w_i);
        return (0.25f * f * d * g) / (cos_o * cos_i);
    }
    // ---- DisneyGlass :779-825
    Color evaluate_glass(const Material &m, float mat_eta, V3 w_o, V3 h, V3 w_i, const HitRecord &rec) const {
        float eta = rec.front_face ? 1.0f / mat_eta : mat_eta;
        bool refl = w_o.y * w_i.y > 0.0f;
        float cos_o = rmin(std::fabs(w_o.y), 1.0f);
        float cos_i = rmin(std::fabs(w_i.y), 1.0f);
        float h_dot_o = rclamp(dot(h, w_o), -1.0f, 1.0f);
        float h_dot_i = rclamp(dot(h, w_i), -1.0f, 1.0f);
        Color base_color = tex(m, rec);
        float f = fresnel_o(w_o, h, eta);
        float d = m.mf.ggx_glass(h);
        float g = m.mf.separable_g(w_o) * m.mf.separable_g(w_i);
        if (refl) return ((0.25f * f * d * g) / (cos_o * cos_i)) * base_color;
        float sqrt_denom = h_dot_o + eta * h_dot_i;
        Color bc = v3(std::sqrt(base_color.x), std::sqrt(base_color.y), std::sqrt(base_color.z));
        float val = ((1.0f - f) * d * g * std::fabs(h_dot_i * h_dot_o)) / (cos_o * cos_i * sqrt_denom * sqrt_denom);
        return val * bc;
    }
    V3 scatter_glass(const Material &m, float mat_eta, Rng &rng, V3 w_out, const HitRecord &rec) const {  // :857-921
        V3 n_e = m.mf.sample_ne(rng, w_out);
        float eta = rec.front_face ? 1.0f / mat_eta : mat_eta;
        float u3 = rng.range(0.0f, 1.0f);
        float fresnel = fresnel_o(w_out, n_e, eta);
        if (fresnel >= u3) return reflect(w_out, n_e);
        V3 w_in;
        if (refract(w_out, n_e, eta, w_in)) return w_in;
        return reflect(w_out, n_e);
    }
    float pdf_glass(const Material &m, float mat_eta, V3 w_out, V3 w_in, const HitRecord &rec) const {  // :923-953
        float eta = rec.front_face ? 1.0f / mat_eta : mat_eta;
        if (w_out.y * w_in.y > 0.0f) {
            V3 h = normalize(w_out + w_in);
            float d = m.mf.ggx_glass(h);
            float g_o = m.mf.separable_g(w_out);
            float cos_o = rmin(std::fabs(w_out.y), 1.0f);
            float fresnel = fresnel_o(w_out, h, eta);
            return (0.25f * fresnel * d * g_o) / cos_o;
        }
        V3 h = normalize(eta * w_out + w_in);
        if (h.y < 0.0f) h = -h;
        float fresnel = fresnel_o(w_out, h, eta);
        float d = m.mf.ggx_glass(h);
        float g_o = m.mf.separable_g(w_out);
        float h_dot_i = dot(h, w_in);
        float h_dot_o = dot(h, w_out);
        float sqrt_denom = h_dot_o + eta * h_dot_i;
        float dh_di = eta * eta * h_dot_i / (sqrt_denom * sqrt_denom);
        float cos_o = rmin(std::fabs(w_out.y), 1.0f);
        return (1.0f - fresnel) * d * g_o * std::fabs(dh_di * h_dot_o / (cos_o));
    }
    float pdf_metal(const Material &m, V3 w_out, V3 w_in) const {  // :694-701
        V3 h = normalize(w_out + w_in);
        float d = m.mf.ggx_metal(h);
        float g_o = m.mf.separable_g(w_out);
        float cos_i = rmin(std::fabs(w_in.y), 1.0f);
        return (0.25f * d * g_o) / cos_i;
    }
    V3 scatter_clearcoat(const Material &m, Rng &rng, V3 w_out) const {  // :482-508
        float a2 = m.cc_a2;
        float u_0 = rng.range(0.0f, 1.0f);
        float u_1 = rng.range(0.0f, 1.0f);
        float cos_elevation2 = (1.0f - std::pow(a2, 1.0f - u_0)) / (1.0f - a2);
        float sin_elevation = std::sqrt(1.0f - cos_elevation2);
        float cos_elevation = std::sqrt(cos_elevation2);
        float azimuth = 2.0f * PI_F * u_1;
        float sin_azimuth = std::sin(azimuth);
        float cos_azimuth = std::cos(azimuth);
        V3 n_e = v3(sin_elevation * cos_azimuth, cos_elevation, sin_elevation * sin_azimuth);
        return reflect(w_out, n_e);
    }
    float pdf_clearcoat(const Material &m, V3 w_out, V3 w_in) const {  // :510-519
        float a2 = m.cc_a2;
        V3 h = normalize(w_out + w_in);
        float cos_d = rmin(std::fabs(dot(h, w_in)), 1.0f);
        float cos_h = rmin(std::fabs(h.y), 1.0f);
        float d = gtr1(m, cos_d, a2);
        return (d * cos_h) / (4.0f * cos_d);
    }

    // ---- combined Disney :154-303
    struct DisneyWeights { float diffuse, metal, glass, clearcoat, total; };
    static DisneyWeights disney_weights(const Material &m, const HitRecord &rec) {
        float metallic = m.p[1], spec_trans = m.p[2], clearcoat = m.p[9];
        bool transmitted = !rec.front_face;
        DisneyWeights w;
        w.diffuse = transmitted ? 0.0f : (1.0f - metallic) * (1.0f - spec_trans);
        w.metal = transmitted ? 0.0f : 1.0f - spec_trans * (1.0f - metallic);
        w.glass = (1.0f - metallic) * spec_trans;
        w.clearcoat = transmitted ? 0.0f : 0.25f * clearcoat;
        w.total = w.diffuse + w.metal + w.glass + w.clearcoat;
        return w;
    }
    Color evaluate_disney(const Material &m, V3 w_o, V3 h, V3 w_i, const HitRecord &rec) const {
        float subsurface = m.p[0], metallic = m.p[1], spec_trans = m.p[2], specular = m.p[3], specular_tint = m.p[4];
        float roughness = m.p[5], sheen = m.p[7], sheen_tint = m.p[8], clearcoat = m.p[9], ior = m.p[11];
        bool transmitted = !rec.front_face;
        Color f_diffuse = transmitted ? gray(0.0f) : evaluate_diffuse(m, roughness, subsurface, w_o, h, w_i, rec);
        Color f_sheen = transmitted ? gray(0.0f) : evaluate_sheen(m, sheen, sheen_tint, h, w_i, rec);
        Color f_metal = gray(0.0f);
        if (!transmitted) {
            Color base_color = tex(m, rec);
            V3 c_tint = calculate_tint(base_color);
            V3 k_s = (1.0f - specular_tint) * v3(1, 1, 1) + specular_tint * c_tint;
            Color c_0 = specular * r_0(ior) * (1.0f - metallic) * k_s + metallic * base_color;
            f_metal = evaluate_metal_color(m, w_o, h, w_i, c_0);
        }
        Color f_clearcoat = transmitted ? gray(0.0f) : evaluate_clearcoat(m, clearcoat, w_o, h, w_i);
        Color f_glass = evaluate_glass(m, ior, w_o, h, w_i, rec);
        Color diffuse = (1.0f - spec_trans) * (1.0f - metallic) * f_diffuse;
        Color sheen_c = (1.0f - metallic) * sheen * f_sheen;
        Color metal = (1.0f - (spec_trans * (1.0f - metallic))) * f_metal;
        Color clearcoat_c = 0.25f * clearcoat * f_clearcoat;
        Color glass = (1.0f - metallic) * spec_trans * f_glass;
        return diffuse + sheen_c + metal + clearcoat_c + glass;
    }

    // ------------------------------------------------- the Material trait
    V3 scatter(const Material &m, Rng &rng, V3 w_out, const HitRecord &rec) const {
        switch (m.type) {
            case PTS_MAT_LAMBERTIAN:
            case PTS_MAT_DISNEY_DIFFUSE:
            case PTS_MAT_DISNEY_SHEEN:
                return cosine_sample(rng);
            case PTS_MAT_METAL: {  // metal.rs:31-47
                V3 fuzz = gray(0.0f);
                if (m.p[3] > 0.0f) fuzz = m.p[3] * normalize(sphere_pdf_generate(rng));
                return reflect_y(-normalize(w_out)) + fuzz;
            }
            case PTS_MAT_DIELECTRIC: {  // dielectric.rs:34-57
                float ri = rec.front_face ? 1.0f / m.p[0] : m.p[0];
                V3 unit_direction = -normalize(w_out);
                float cos_theta = -rmin(unit_direction.y, 1.0f);
                float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
                float reflect_thresh = rng.range(0.0f, 1.0f);
                bool cannot_refract = ri * sin_theta > 1.0f;
                float fresnel = fresnel_o(unit_direction, v3(0, 1, 0), ri);
                if (cannot_refract || fresnel > reflect_thresh) return reflect_y(unit_direction);
                return refract_y(unit_direction, ri);
            }
            case PTS_MAT_DISNEY: {  // disney.rs:232-271
                DisneyWeights w = disney_weights(m, rec);
                float u1 = rng.range(0.0f, w.total);
                if (u1 < w.diffuse) return cosine_sample(rng);
                if (u1 - w.diffuse < w.metal) return reflect(w_out, m.mf.sample_ne(rng, w_out));
                if (u1 - (w.diffuse + w.metal) < w.glass) return scatter_glass(m, m.p[11], rng, w_out, rec);
                return scatter_clearcoat(m, rng, w_out);
            }
            case PTS_MAT_DISNEY_METAL: return reflect(w_out, m.mf.sample_ne(rng, w_out));  // :651-692
            case PTS_MAT_DISNEY_GLASS: return scatter_glass(m, m.p[2], rng, w_out, rec);
            case PTS_MAT_DISNEY_CLEARCOAT: return scatter_clearcoat(m, rng, w_out);
            default: return gray(0.0f);  // trait default, materials.rs:89-98
        }
    }
    Color bsdf_evaluate(const Material &m, V3 w_out, V3 w_in, const HitRecord &rec) const {
        switch (m.type) {
            case PTS_MAT_LAMBERTIAN: return tex(m, rec) / PI_F;             // lambertian.rs:60-62
            case PTS_MAT_METAL: return v3(m.p[0], m.p[1], m.p[2]);          // metal.rs:49-56
            case PTS_MAT_DIELECTRIC: return gray(1.0f);                     // dielectric.rs:59-66
            case PTS_MAT_DISNEY: {                                          // disney.rs:215-230
                float eta = rec.front_face ? 1.0f / m.p[11] : m.p[11];
                bool refl = w_out.y * w_in.y > 0.0f;
                V3 h = refl ? normalize(w_out + w_in) : normalize(eta * w_out + w_in);
                return evaluate_disney(m, w_out, h, w_in, rec);
            }
            case PTS_MAT_DISNEY_DIFFUSE: {  // :1064-1067
                V3 h = normalize(w_out + w_in);
                return evaluate_diffuse(m, m.p[0], m.p[1], w_out, h, w_in, rec);
            }
            case PTS_MAT_DISNEY_METAL: {  // :643-649
                if (w_in.y < 0.0f) return gray(0.0f);
                V3 h = normalize(w_out + w_in);
                return evaluate_metal_color(m, w_out, h, w_in, tex(m, rec));
            }
            case PTS_MAT_DISNEY_GLASS: {  // :837-855
                float eta = rec.front_face ? 1.0f / m.p[2] : m.p[2];
                bool refl = w_out.y * w_in.y > 0.0f;
                V3 h = refl ? normalize(w_out + w_in) : normalize(eta * w_out + w_in);
                if (h.y < 0.0f) h = -h;
                return evaluate_glass(m, m.p[2], w_out, h, w_in, rec);
            }
            case PTS_MAT_DISNEY_CLEARCOAT: {  // :477-480
                V3 h = normalize(w_out + w_in);
                return evaluate_clearcoat(m, m.p[0], w_out, h, w_in);
            }
            case PTS_MAT_DISNEY_SHEEN: {  // :383-386
                V3 h = normalize(w_out + w_in);
                return evaluate_sheen(m, m.p[0], m.p[1], h, w_in, rec);
            }
            default: return gray(0.0f);
        }
    }
    float scattering_pdf(const Material &m, V3 w_out, V3 w_in, const HitRecord &rec) const {
        switch (m.type) {
            case PTS_MAT_LAMBERTIAN: return rclamp(w_in.y, 0.0f, 1.0f) / PI_F;  // lambertian.rs:64-66
            case PTS_MAT_DISNEY: {                                               // disney.rs:273-303
                DisneyWeights w = disney_weights(m, rec);
                float pdf_diffuse = w.diffuse * (w_in.y * INV_PI);
                float pdf_metal_ = w.metal * pdf_metal(m, w_out, w_in);
                float pdf_glass_ = w.glass * pdf_glass(m, m.p[11], w_out, w_in, rec);
                float pdf_clearcoat_ = w.clearcoat * pdf_clearcoat(m, w_out, w_in);
                return (pdf_diffuse + pdf_metal_ + pdf_glass_ + pdf_clearcoat_) / w.total;
            }
            case PTS_MAT_DISNEY_DIFFUSE:
            case PTS_MAT_DISNEY_SHEEN: return w_in.y * INV_PI;  // :1069-1071, :388-390
            case PTS_MAT_DISNEY_METAL: return pdf_metal(m, w_out, w_in);
            case PTS_MAT_DISNEY_GLASS: return pdf_glass(m, m.p[2], w_out, w_in, rec);
            case PTS_MAT_DISNEY_CLEARCOAT: return pdf_clearcoat(m, w_out, w_in);
            default: return 0.0f;  // trait default, materials.rs:120-122
        }
    }
    Color emitted(const Material &m, const HitRecord &rec) const {  // light.rs:42-47
        if (m.type != PTS_MAT_DIFFUSE_LIGHT) return gray(0.0f);
        if (!rec.front_face) return gray(0.0f);
        return tt->value(m.texture, rec.u, rec.v, rec.p);
    }
};

}  // namespace oracle
