// Device-side shading: hit-record reconstruction, textures, the ten material types,
// light sampling / light pdf.  Local shading frame is y-up, w_out.y >= 0.
//
// The formulas are the reference's, quirks included (they are what parity is judged
// against); the reference line ranges are given at each function.
#pragma once
#include "dev_traverse.h"

namespace ptd {

// What HitRecord carries (src/objects.rs:14-27), rebuilt from the compact hit.
struct Surf {
    f3 p, normal;
    float t;
    bool front_face;
    int32_t material;
    float u, v;
};

// ---------------------------------------------------------------- textures
// PTS_TEX_NOISE, which the reference does not have (its README names the "Next Week" scene, whose
// marble texture this is): gradient noise on the integer lattice.  The gradient of a lattice point
// is the normalised vector of three 21-bit fields of mix64(seed, point), each mapped to [-1, 1).
__device__ __forceinline__ int32_t f2i32_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}
__device__ inline f3 noise_gradient(uint32_t seed, int32_t ix, int32_t iy, int32_t iz) {
    uint64_t z = (uint64_t)seed ^ ((uint64_t)(uint32_t)ix * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(uint32_t)iy * 0xC2B2AE3D27D4EB4Full) ^
                 ((uint64_t)(uint32_t)iz * 0x165667B19E3779F9ull);
    z = mix64(z);
    f3 g = mk3((float)(uint32_t)(z & 0x1FFFFFull) / 1048576.0f - 1.0f, (float)(uint32_t)((z >> 21) & 0x1FFFFFull) / 1048576.0f - 1.0f,
               (float)(uint32_t)((z >> 42) & 0x1FFFFFull) / 1048576.0f - 1.0f);
    const float n2 = dot(g, g);
    if (!(n2 > 0.0f)) return mk3(1.0f, 0.0f, 0.0f);
    return g / sqrtf(n2);
}
__device__ inline float noise_value(uint32_t seed, f3 p) {
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const int32_t ix = f2i32_sat(fx), iy = f2i32_sat(fy), iz = f2i32_sat(fz);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const float su = u * u * (3.0f - 2.0f * u), sv = v * v * (3.0f - 2.0f * v), sw = w * w * (3.0f - 2.0f * w);
    float acc = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                const f3 g = noise_gradient(seed, (int32_t)((uint32_t)ix + (uint32_t)di), (int32_t)((uint32_t)iy + (uint32_t)dj), (int32_t)((uint32_t)iz + (uint32_t)dk));
                const f3 d = mk3(u - (float)di, v - (float)dj, w - (float)dk);
                const float weight = ((di ? su : 1.0f - su) * (dj ? sv : 1.0f - sv)) * (dk ? sw : 1.0f - sw);
                acc += weight * dot(g, d);
            }
    return acc;
}
__device__ inline float noise_turbulence(uint32_t seed, f3 p) {
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; i++) {
        accum += weight * noise_value(seed, p);
        weight *= 0.5f;
        p = p * 2.0f;
    }
    return fabsf(accum);
}

// Texture::value (src/textures.rs:59-103, src/textures/image.rs:31-65). Checkered nests.
__device__ inline f3 tex_value(const DevScene &sc, int32_t id, float u, float v, f3 p) {
    for (;;) {
        const pts_texture t = sc.textures[id];
        if (t.type == PTS_TEX_SOLID) return mk3(t.color[0], t.color[1], t.color[2]);
        if (t.type == PTS_TEX_NOISE) {
            const float k = 1.0f + sinf(t.scale * p.z + 10.0f * noise_turbulence((uint32_t)t.odd, p));
            return mk3(t.color[0], t.color[1], t.color[2]) * k;
        }
        if (t.type == PTS_TEX_CHECKERED) {
            long long ui = f2i64_sat(floorf(u * t.scale));
            long long vi = f2i64_sat(floorf(v * t.scale));
            id = ((ui + vi) % 2 == 0) ? t.even : t.odd;
            continue;
        }
        // PTS_TEX_IMAGE
        if (t.image < 0) {
            long long ui = f2i64_sat(floorf(u * 64.0f));
            long long vi = f2i64_sat(floorf(v * 64.0f));
            float g = ((ui + vi) % 2 == 0) ? 1.0f : 0.7f;
            return mk3(g, g, g);
        }
        const pts_image im = sc.images[t.image];
        if (im.height == 0 || im.width == 0) return mk3(0.0f, 0.0f, 0.0f);
        float uu = rclamp(u, 0.0f, 1.0f);
        float vv = 1.0f - rclamp(v, 0.0f, 1.0f);
        uint32_t i = f2u32_sat(uu * (float)(im.width - 1));
        uint32_t j = f2u32_sat(vv * (float)(im.height - 1));
        const uint8_t *px = sc.texels + im.offset + 3ull * ((uint64_t)j * im.width + i);
        return mk3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
}

// ---------------------------------------------------------------- hit record
// Sphere::get_uv (src/objects/sphere.rs:29-36)
__device__ __forceinline__ void sphere_uv(f3 p, float &u, float &v) {
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PTD_PI;
    u = phi / (2.0f * PTD_PI);
    v = theta / PTD_PI;
}

// Rebuilds the HitRecord of the accepted candidate: Triangle::hit (:199-217),
// Sphere::hit (:82-96), Quad::hit (:137-157), then Instance::hit (:391-397).
// `time`: the path's time, for the spheres that move (ignored where the scene has none).
__device__ inline Surf make_surf(const DevScene &sc, f3 ro, f3 rd, float t, float hu, float hv, uint32_t prim,
                                 uint32_t code, float time = 0.0f) {
    Surf s;
    const uint32_t type = code & 3u, inst1 = code >> 2;
    Affine fwd;
    float4 inst_last = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (inst1) {
        const float4 *ip = sc.insts + 7 * (size_t)(sc.world.insts.prim_first + inst1 - 1u);
        const Affine inv = load_affine(ip);
        fwd = load_affine(ip + 3);
        inst_last = ip[6];
        ro = xf_point(inv, ro);
        rd = xf_vector(inv, rd);
        chain_into(sc, inst_last, ro, rd);
    }
    f3 outward;
    if (type == PT_PRIM_TRI) {
        const float4 *c = sc.tri_cold + 6 * (size_t)prim;
        const float4 c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5];
        // rows of the 8x3 matrix: (c0.x c0.y c0.z) (c0.w c1.x c1.y) (c1.z c1.w c2.x) (c2.y c2.z c2.w)
        //                         (c3.x c3.y c3.z) (c3.w c4.x c4.y) (c4.z c4.w c5.x) (c5.y c5.z c5.w)
        const float w0 = 1.0f - hu - hv, w1 = hu, w2 = hv;
#define PTD_ROW(a, b, cc) (((a)*w0 + (b)*w1) + (cc)*w2)
        s.p = mk3(PTD_ROW(c0.x, c0.y, c0.z), PTD_ROW(c0.w, c1.x, c1.y), PTD_ROW(c1.z, c1.w, c2.x));
        outward = normalize(mk3(PTD_ROW(c2.y, c2.z, c2.w), PTD_ROW(c3.x, c3.y, c3.z), PTD_ROW(c3.w, c4.x, c4.y)));
        s.u = PTD_ROW(c4.z, c4.w, c5.x);
        s.v = PTD_ROW(c5.y, c5.z, c5.w);
#undef PTD_ROW
        s.material = sc.tri_mat[prim];
    } else if (type == PT_PRIM_SPHERE) {
        const float4 sp = sc.spheres[prim];
        f3 at = ro + t * rd;
        f3 center = mk3(sp.x, sp.y, sp.z);
        if (sc.n_moving != 0u && prim >= sc.n_static_spheres) {
            const float4 dl = sc.moving_delta[prim - sc.n_static_spheres];
            center = center + time * mk3(dl.x, dl.y, dl.z);
        }
        outward = (at - center) / sp.w;
        s.p = at;
        sphere_uv(outward, s.u, s.v);
        s.material = sc.sphere_mat[prim];
    } else {
        const float4 q2 = sc.quads[4 * (size_t)prim + 2];
        outward = mk3(q2.y, q2.z, q2.w);
        s.p = ro + t * rd;
        s.u = hu;
        s.v = hv;
        s.material = sc.quad_mat[prim];
    }
    s.t = t;
    s.front_face = dot(rd, outward) < 0.0f;  // set_face_normal, src/objects.rs:82-90
    s.normal = s.front_face ? outward : -outward;
    if (inst1) {
        chain_out_of(sc, inst_last, s.p, s.normal);
        s.p = xf_point(fwd, s.p);
        s.normal = xf_vector(fwd, s.normal);
    }
    return s;
}

// ---------------------------------------------------------------- helpers (src/materials.rs)
__device__ __forceinline__ f3 reflect(f3 v, f3 n) {  // :16-19
    f3 m = -v;
    return m - 2.0f * dot(m, n) * n;
}
__device__ __forceinline__ f3 reflect_y(f3 v) { return mk3(v.x, -v.y, v.z); }
__device__ __forceinline__ bool refract(f3 w_o, f3 n, float eta, f3 &out) {  // :28-40
    float cos_o = rmin(dot(w_o, n), 1.0f);
    float sin2_o = 1.0f - cos_o * cos_o;
    float sin2_i = sin2_o * eta * eta;
    if (sin2_i >= 1.0f) return false;
    float cos_i = sqrtf(1.0f - sin2_i);
    out = normalize((eta * -w_o) + ((eta * cos_o) - cos_i) * n);
    return true;
}
__device__ __forceinline__ f3 refract_y(f3 uv, float etai_over_etat) {  // :68-74
    const f3 up = mk3(0.0f, 1.0f, 0.0f);
    float cos_theta = -rmin(uv.y, 1.0f);
    f3 r_out_perp = etai_over_etat * (uv + (cos_theta * up));
    f3 r_out_parallel = -sqrtf(fabsf(1.0f - norm_squared(r_out_perp))) * up;
    return r_out_perp + r_out_parallel;
}
// src/materials/disney.rs:26-74
__device__ __forceinline__ float r_0(float ior) { return ((ior - 1.0f) * (ior - 1.0f)) / ((ior + 1.0f) * (ior + 1.0f)); }
__device__ __forceinline__ float schlick_weight(float x) { return powf(1.0f - x, 5.0f); }
__device__ __forceinline__ float schlick(float ior, float x) {
    float r0 = r_0(ior);
    return r0 + (1.0f - r0) * schlick_weight(x);
}
__device__ __forceinline__ float fresnel_o(f3 w_o, f3 n, float eta) {
    float cos_o = dot(w_o, n);
    float sin2_o = rmax(1.0f - cos_o * cos_o, 0.0f);
    float sin2_i = sin2_o * eta * eta;
    if (sin2_i >= 1.0f) return 1.0f;
    cos_o = fabsf(cos_o);
    float cos_i = sqrtf(rmax(1.0f - sin2_i, 0.0f));
    eta = 1.0f / eta;
    float r_para = (cos_o - eta * cos_i) / (cos_o + eta * cos_i);
    float r_perp = (eta * cos_o - cos_i) / (eta * cos_o + eta * cos_i);
    return 0.5f * (r_para * r_para + r_perp * r_perp);
}
__device__ __forceinline__ f3 calculate_tint(f3 base_color) {
    float luminance = rclamp(dot(mk3(0.3f, 0.6f, 1.0f), base_color), 0.0f, 1.0f);
    if (luminance > 0.0f) return (1.0f / luminance) * base_color;
    return mk3(1.0f, 1.0f, 1.0f);
}

// Cosine-weighted hemisphere sample: lambertian.rs:47-57, disney.rs:370-380, :1051-1061.
__device__ __forceinline__ f3 sample_cosine(Rng &rng) {
    float r1 = rng.range(0.0f, 1.0f);
    float r2 = rng.range(0.0f, 1.0f);
    float phi = 2.0f * PTD_PI * r1;
    float x = cosf(phi) * sqrtf(r2);
    float z = sinf(phi) * sqrtf(r2);
    float y = sqrtf(1.0f - r2);
    return mk3(x, y, z);
}
// SpherePDF::generate (src/types/pdf.rs:24-36) — rejection loop kept as is.
__device__ __forceinline__ f3 sample_sphere_pdf(Rng &rng) {
    for (;;) {
        float u = rng.range(0.0f, 1.0f), v = rng.range(0.0f, 1.0f);
        float ns = u * u + v * v;
        if (ns <= 1.0f) {
            float x = 2.0f * u * sqrtf(1.0f - ns);
            float y = 2.0f * v * sqrtf(1.0f - ns);
            float z = 1.0f - (2.0f * ns);
            return mk3(x, y, z);
        }
    }
}

// ---------------------------------------------------------------- microfacet pieces (disney.rs)
struct Alpha {
    float a_x, a_z, inv_a_x, inv_a_z;
};
__device__ __forceinline__ Alpha mat_alpha(const DevMaterial &m) { return Alpha{m.derived[0], m.derived[1], m.derived[2], m.derived[3]}; }
__device__ __forceinline__ float ggx_metal(const Alpha &al, f3 h) {  // :575-580
    float k = (h.x * h.x * al.inv_a_x * al.inv_a_x) + (h.z * h.z * al.inv_a_z * al.inv_a_z) + h.y * h.y;
    return PTD_INV_PI * al.inv_a_x * al.inv_a_z * (1.0f / (k * k));
}
__device__ __forceinline__ float ggx_glass(const Alpha &al, f3 h) {  // :765-769
    f3 h2 = cmul(h, h);
    float k = (h2.x * al.inv_a_x * al.inv_a_x) + (h2.z * al.inv_a_z * al.inv_a_z) + h2.y;
    return PTD_INV_PI * al.inv_a_x * al.inv_a_z * (1.0f / (k * k));
}
__device__ __forceinline__ float smith_g1(const Alpha &al, f3 w) {  // :582-588, :771-777
    float x = al.a_x * w.x, y = w.y, z = al.a_z * w.z;
    return 2.0f / (1.0f + sqrtf(1.0f + ((x * x) + (z * z)) / (y * y)));
}
// Heitz VNDF micro-normal (:660-688 = :866-894)
__device__ inline f3 sample_vndf(const Alpha &al, Rng &rng, f3 w_out) {
    float u1 = rng.range(0.0f, 1.0f);
    float u2 = rng.range(0.0f, 1.0f);
    f3 w_hemi = normalize(mk3(al.a_x * w_out.x, w_out.y, al.a_z * w_out.z));
    f3 basis_1 = (w_hemi.y < 0.9999f) ? normalize(cross(mk3(0.0f, 1.0f, 0.0f), w_hemi)) : mk3(1.0f, 0.0f, 0.0f);
    f3 basis_2 = cross(w_hemi, basis_1);
    float r = sqrtf(u1);
    float phi = 2.0f * PTD_PI * u2;
    float t1 = r * cosf(phi);
    float t2 = r * sinf(phi);
    float s = 0.5f * (1.0f + w_hemi.y);
    t2 = (1.0f - s) * sqrtf(1.0f - t1 * t1) + (s * t2);
    f3 n_h = (t1 * basis_1) + (t2 * basis_2) + sqrtf(rmax(1.0f - (t1 * t1) - (t2 * t2), 0.0f)) * w_hemi;
    return normalize(mk3(al.a_x * n_h.x, rmax(n_h.y, 0.0f), al.a_z * n_h.z));
}

// DisneyDiffuse::evaluate_diffuse :989-1032
__device__ inline f3 eval_diffuse(float roughness, float subsurface, f3 albedo, f3 w_o, f3 h, f3 w_i) {
    float cos_o = rmin(fabsf(w_o.y), 1.0f);
    float cos_i = rmin(fabsf(w_i.y), 1.0f);
    float cos_d = rmin(fabsf(dot(h, w_i)), 1.0f);
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
    return f_diffuse * albedo * PTD_INV_PI;
}
// DisneySheen::evaluate_sheen :331-351
__device__ inline f3 eval_sheen(float sheen, float sheen_tint, f3 base_color, f3 h, f3 w_i) {
    if (sheen <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float h_dot_w_i = dot(h, w_i);
    f3 tint = calculate_tint(base_color);
    return sheen * lerp3(mk3(1.0f, 1.0f, 1.0f), tint, sheen_tint) * schlick_weight(h_dot_w_i);
}
// DisneyClearcoat :415-465
__device__ __forceinline__ float gtr1(float a2, float a2_invlog2, float cos_d) {
    if (a2 >= 1.0f) return PTD_INV_PI;
    return PTD_INV_PI * (a2 - 1.0f) * a2_invlog2 / (1.0f + (a2 - 1.0f) * cos_d * cos_d);
}
__device__ __forceinline__ float clearcoat_g(f3 w) {
    float x = 0.25f * w.x, y = w.y, z = 0.25f * w.z;
    return 2.0f / sqrtf(1.0f + (((x * x) + (z * z)) / (y * y)));
}
__device__ inline f3 eval_clearcoat(float clearcoat, float a2, float a2_invlog2, f3 w_o, f3 h, f3 w_i) {
    if (clearcoat <= 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    float cos_o = rmin(fabsf(w_o.y), 1.0f);
    float cos_i = rmin(fabsf(w_i.y), 1.0f);
    float cos_d = rmin(fabsf(dot(h, w_i)), 1.0f);
    float d = gtr1(a2, a2_invlog2, cos_d);
    float f = schlick(1.5f, cos_d);
    float g = clearcoat_g(w_i) * clearcoat_g(w_o);
    float value = (0.25f * clearcoat * d * f * g) / (cos_o * cos_i);
    return mk3(value, value, value);
}
__device__ inline f3 sample_clearcoat(float a2, Rng &rng, f3 w_out) {  // :482-508
    float u_0 = rng.range(0.0f, 1.0f);
    float u_1 = rng.range(0.0f, 1.0f);
    float cos_elevation2 = (1.0f - powf(a2, 1.0f - u_0)) / (1.0f - a2);
    float sin_elevation = sqrtf(1.0f - cos_elevation2);
    float cos_elevation = sqrtf(cos_elevation2);
    float azimuth = 2.0f * PTD_PI * u_1;
    float sin_azimuth = sinf(azimuth);
    float cos_azimuth = cosf(azimuth);
    f3 n_e = mk3(sin_elevation * cos_azimuth, cos_elevation, sin_elevation * sin_azimuth);
    return reflect(w_out, n_e);
}
__device__ inline float pdf_clearcoat(float a2, float a2_invlog2, f3 w_out, f3 w_in) {  // :510-519
    f3 h = normalize(w_out + w_in);
    float cos_d = rmin(fabsf(dot(h, w_in)), 1.0f);
    float cos_h = rmin(fabsf(h.y), 1.0f);
    float d = gtr1(a2, a2_invlog2, cos_d);
    return (d * cos_h) / (4.0f * cos_d);
}
// DisneyMetal::evaluate_metal / evaluate_metal_achromatic :590-631
__device__ inline f3 eval_metal(const Alpha &al, f3 base_color, f3 w_o, f3 h, f3 w_i) {
    float cos_o = rmin(fabsf(w_o.y), 1.0f);
    float cos_i = rmin(fabsf(w_i.y), 1.0f);
    float cos_d = rmin(fabsf(dot(h, w_i)), 1.0f);
    f3 f = lerp3(base_color, mk3(1.0f, 1.0f, 1.0f), schlick_weight(cos_d));
    float d = ggx_metal(al, h);
    float g = smith_g1(al, w_o) * smith_g1(al, w_i);
    return (0.25f * f * d * g) / (cos_o * cos_i);
}
__device__ inline float pdf_metal(const Alpha &al, f3 w_out, f3 w_in) {  // :694-701
    f3 h = normalize(w_out + w_in);
    float d = ggx_metal(al, h);
    float g_o = smith_g1(al, w_out);
    float cos_i = rmin(fabsf(w_in.y), 1.0f);
    return (0.25f * d * g_o) / cos_i;
}
// DisneyGlass :779-953
__device__ inline f3 eval_glass(const Alpha &al, float mat_eta, f3 base_color, bool front_face, f3 w_o, f3 h, f3 w_i) {
    float eta = front_face ? 1.0f / mat_eta : mat_eta;
    bool refl = w_o.y * w_i.y > 0.0f;
    float cos_o = rmin(fabsf(w_o.y), 1.0f);
    float cos_i = rmin(fabsf(w_i.y), 1.0f);
    float h_dot_o = rclamp(dot(h, w_o), -1.0f, 1.0f);
    float h_dot_i = rclamp(dot(h, w_i), -1.0f, 1.0f);
    float f = fresnel_o(w_o, h, eta);
    float d = ggx_glass(al, h);
    float g = smith_g1(al, w_o) * smith_g1(al, w_i);
    if (refl) return ((0.25f * f * d * g) / (cos_o * cos_i)) * base_color;
    float sqrt_denom = h_dot_o + eta * h_dot_i;
    f3 bc = mk3(sqrtf(base_color.x), sqrtf(base_color.y), sqrtf(base_color.z));
    float val = ((1.0f - f) * d * g * fabsf(h_dot_i * h_dot_o)) / (cos_o * cos_i * sqrt_denom * sqrt_denom);
    return val * bc;
}
__device__ inline f3 sample_glass(const Alpha &al, float mat_eta, bool front_face, Rng &rng, f3 w_out) {  // :857-921
    f3 n_e = sample_vndf(al, rng, w_out);
    float eta = front_face ? 1.0f / mat_eta : mat_eta;
    float u3 = rng.range(0.0f, 1.0f);
    float fresnel = fresnel_o(w_out, n_e, eta);
    if (fresnel >= u3) return reflect(w_out, n_e);
    f3 w_in;
    if (refract(w_out, n_e, eta, w_in)) return w_in;
    return reflect(w_out, n_e);
}
__device__ inline float pdf_glass(const Alpha &al, float mat_eta, bool front_face, f3 w_out, f3 w_in) {  // :923-953
    float eta = front_face ? 1.0f / mat_eta : mat_eta;
    if (w_out.y * w_in.y > 0.0f) {
        f3 h = normalize(w_out + w_in);
        float d = ggx_glass(al, h);
        float g_o = smith_g1(al, w_out);
        float cos_o = rmin(fabsf(w_out.y), 1.0f);
        float fresnel = fresnel_o(w_out, h, eta);
        return (0.25f * fresnel * d * g_o) / cos_o;
    }
    f3 h = normalize(eta * w_out + w_in);
    if (h.y < 0.0f) h = -h;
    float fresnel = fresnel_o(w_out, h, eta);
    float d = ggx_glass(al, h);
    float g_o = smith_g1(al, w_out);
    float h_dot_i = dot(h, w_in);
    float h_dot_o = dot(h, w_out);
    float sqrt_denom = h_dot_o + eta * h_dot_i;
    float dh_di = eta * eta * h_dot_i / (sqrt_denom * sqrt_denom);
    float cos_o = rmin(fabsf(w_out.y), 1.0f);
    return (1.0f - fresnel) * d * g_o * fabsf(dh_di * h_dot_o / (cos_o));
}

// Lobe weights of the combined Disney material (:232-254 = :274-294)
struct LobeWeights {
    float diffuse, metal, glass, clearcoat, total;
};
__device__ __forceinline__ LobeWeights disney_weights(const DevMaterial &m, bool front_face) {
    const float metallic = m.p[1], spec_trans = m.p[2], clearcoat = m.p[9];
    const bool transmitted = !front_face;
    LobeWeights w;
    w.diffuse = transmitted ? 0.0f : (1.0f - metallic) * (1.0f - spec_trans);
    w.metal = transmitted ? 0.0f : 1.0f - spec_trans * (1.0f - metallic);
    w.glass = (1.0f - metallic) * spec_trans;
    w.clearcoat = transmitted ? 0.0f : 0.25f * clearcoat;
    w.total = w.diffuse + w.metal + w.glass + w.clearcoat;
    return w;
}

// ---------------------------------------------------------------- the Material trait, per type
// MT is a compile-time material type so that each shade kernel carries only its own code.
template <int MT>
__device__ inline f3 mat_scatter(const DevScene &sc, const DevMaterial &m, Rng &rng, f3 w_out, const Surf &s) {
    (void)sc;
    if constexpr (MT == PTS_MAT_LAMBERTIAN || MT == PTS_MAT_DISNEY_DIFFUSE || MT == PTS_MAT_DISNEY_SHEEN) {
        return sample_cosine(rng);
    } else if constexpr (MT == PTS_MAT_METAL) {  // metal.rs:31-47
        f3 fuzz = mk3(0.0f, 0.0f, 0.0f);
        if (m.p[3] > 0.0f) fuzz = m.p[3] * normalize(sample_sphere_pdf(rng));
        return reflect_y(-normalize(w_out)) + fuzz;
    } else if constexpr (MT == PTS_MAT_DIELECTRIC) {  // dielectric.rs:34-57
        float ri = s.front_face ? 1.0f / m.p[0] : m.p[0];
        f3 unit_direction = -normalize(w_out);
        float cos_theta = -rmin(unit_direction.y, 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float reflect_thresh = rng.range(0.0f, 1.0f);
        bool cannot_refract = ri * sin_theta > 1.0f;
        float fresnel = fresnel_o(unit_direction, mk3(0.0f, 1.0f, 0.0f), ri);
        if (cannot_refract || fresnel > reflect_thresh) return reflect_y(unit_direction);
        return refract_y(unit_direction, ri);
    } else if constexpr (MT == PTS_MAT_DISNEY) {  // disney.rs:232-271
        const LobeWeights w = disney_weights(m, s.front_face);
        const Alpha al = mat_alpha(m);
        float u1 = rng.range(0.0f, w.total);
        if (u1 < w.diffuse) return sample_cosine(rng);
        if (u1 - w.diffuse < w.metal) return reflect(w_out, sample_vndf(al, rng, w_out));
        if (u1 - (w.diffuse + w.metal) < w.glass) return sample_glass(al, m.p[11], s.front_face, rng, w_out);
        return sample_clearcoat(m.derived[4], rng, w_out);
    } else if constexpr (MT == PTS_MAT_DISNEY_METAL) {
        return reflect(w_out, sample_vndf(mat_alpha(m), rng, w_out));
    } else if constexpr (MT == PTS_MAT_DISNEY_GLASS) {
        return sample_glass(mat_alpha(m), m.p[2], s.front_face, rng, w_out);
    } else if constexpr (MT == PTS_MAT_DISNEY_CLEARCOAT) {
        return sample_clearcoat(m.derived[4], rng, w_out);
    } else {
        return mk3(0.0f, 0.0f, 0.0f);
    }
}

template <int MT>
__device__ inline f3 mat_eval(const DevScene &sc, const DevMaterial &m, f3 w_out, f3 w_in, const Surf &s) {
    if constexpr (MT == PTS_MAT_LAMBERTIAN) {
        return tex_value(sc, m.texture, s.u, s.v, s.p) / PTD_PI;  // lambertian.rs:60-62
    } else if constexpr (MT == PTS_MAT_METAL) {
        return mk3(m.p[0], m.p[1], m.p[2]);
    } else if constexpr (MT == PTS_MAT_DIELECTRIC) {
        return mk3(1.0f, 1.0f, 1.0f);
    } else if constexpr (MT == PTS_MAT_DISNEY) {  // disney.rs:154-230
        const float subsurface = m.p[0], metallic = m.p[1], spec_trans = m.p[2], specular = m.p[3], specular_tint = m.p[4];
        const float roughness = m.p[5], sheen = m.p[7], sheen_tint = m.p[8], clearcoat = m.p[9], ior = m.p[11];
        const Alpha al = mat_alpha(m);
        float eta = s.front_face ? 1.0f / ior : ior;
        bool refl = w_out.y * w_in.y > 0.0f;
        f3 h = refl ? normalize(w_out + w_in) : normalize(eta * w_out + w_in);
        const bool transmitted = !s.front_face;
        const f3 zero = mk3(0.0f, 0.0f, 0.0f);
        const f3 base_color = tex_value(sc, m.texture, s.u, s.v, s.p);
        f3 f_diffuse = transmitted ? zero : eval_diffuse(roughness, subsurface, base_color, w_out, h, w_in);
        f3 f_sheen = transmitted ? zero : eval_sheen(sheen, sheen_tint, base_color, h, w_in);
        f3 f_metal = zero;
        if (!transmitted) {
            f3 c_tint = calculate_tint(base_color);
            f3 k_s = (1.0f - specular_tint) * mk3(1.0f, 1.0f, 1.0f) + specular_tint * c_tint;
            f3 c_0 = specular * r_0(ior) * (1.0f - metallic) * k_s + metallic * base_color;
            f_metal = eval_metal(al, c_0, w_out, h, w_in);
        }
        f3 f_clearcoat = transmitted ? zero : eval_clearcoat(clearcoat, m.derived[4], m.derived[5], w_out, h, w_in);
        f3 f_glass = eval_glass(al, ior, base_color, s.front_face, w_out, h, w_in);
        f3 diffuse = (1.0f - spec_trans) * (1.0f - metallic) * f_diffuse;
        f3 sheen_c = (1.0f - metallic) * sheen * f_sheen;
        f3 metal = (1.0f - (spec_trans * (1.0f - metallic))) * f_metal;
        f3 clearcoat_c = 0.25f * clearcoat * f_clearcoat;
        f3 glass = (1.0f - metallic) * spec_trans * f_glass;
        return diffuse + sheen_c + metal + clearcoat_c + glass;
    } else if constexpr (MT == PTS_MAT_DISNEY_DIFFUSE) {
        f3 h = normalize(w_out + w_in);
        return eval_diffuse(m.p[0], m.p[1], tex_value(sc, m.texture, s.u, s.v, s.p), w_out, h, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_METAL) {
        if (w_in.y < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
        f3 h = normalize(w_out + w_in);
        return eval_metal(mat_alpha(m), tex_value(sc, m.texture, s.u, s.v, s.p), w_out, h, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_GLASS) {
        float eta = s.front_face ? 1.0f / m.p[2] : m.p[2];
        bool refl = w_out.y * w_in.y > 0.0f;
        f3 h = refl ? normalize(w_out + w_in) : normalize(eta * w_out + w_in);
        if (h.y < 0.0f) h = -h;
        return eval_glass(mat_alpha(m), m.p[2], tex_value(sc, m.texture, s.u, s.v, s.p), s.front_face, w_out, h, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_CLEARCOAT) {
        f3 h = normalize(w_out + w_in);
        return eval_clearcoat(m.p[0], m.derived[4], m.derived[5], w_out, h, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_SHEEN) {
        f3 h = normalize(w_out + w_in);
        return eval_sheen(m.p[0], m.p[1], tex_value(sc, m.texture, s.u, s.v, s.p), h, w_in);
    } else {
        return mk3(0.0f, 0.0f, 0.0f);
    }
}

template <int MT>
__device__ inline float mat_pdf(const DevMaterial &m, f3 w_out, f3 w_in, const Surf &s) {
    if constexpr (MT == PTS_MAT_LAMBERTIAN) {
        return rclamp(w_in.y, 0.0f, 1.0f) / PTD_PI;
    } else if constexpr (MT == PTS_MAT_DISNEY) {  // disney.rs:273-303
        const LobeWeights w = disney_weights(m, s.front_face);
        const Alpha al = mat_alpha(m);
        float pdf_diffuse = w.diffuse * (w_in.y * PTD_INV_PI);
        float pdf_m = w.metal * pdf_metal(al, w_out, w_in);
        float pdf_g = w.glass * pdf_glass(al, m.p[11], s.front_face, w_out, w_in);
        float pdf_c = w.clearcoat * pdf_clearcoat(m.derived[4], m.derived[5], w_out, w_in);
        return (pdf_diffuse + pdf_m + pdf_g + pdf_c) / w.total;
    } else if constexpr (MT == PTS_MAT_DISNEY_DIFFUSE || MT == PTS_MAT_DISNEY_SHEEN) {
        return w_in.y * PTD_INV_PI;
    } else if constexpr (MT == PTS_MAT_DISNEY_METAL) {
        return pdf_metal(mat_alpha(m), w_out, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_GLASS) {
        return pdf_glass(mat_alpha(m), m.p[2], s.front_face, w_out, w_in);
    } else if constexpr (MT == PTS_MAT_DISNEY_CLEARCOAT) {
        return pdf_clearcoat(m.derived[4], m.derived[5], w_out, w_in);
    } else {
        return 0.0f;
    }
}

// ---------------------------------------------------------------- participating media (pts_medium; not in the reference)
// Does the ray scatter in a medium before t_closest (its closest surface hit, or infinity)?  Media
// are tried in index order, each against the closest event so far; a medium the ray crosses costs
// one draw.
__device__ inline bool medium_event(const DevScene &sc, Rng &rng, f3 o, f3 d, float t_closest, float &t_event, uint32_t &which) {
    bool any = false;
    float best = t_closest;
    for (uint32_t m = 0; m < sc.n_media; m++) {
        const pts_medium md = sc.media[m];
        const f3 oc = mk3(md.center[0], md.center[1], md.center[2]) - o;
        const float a = norm_squared(d);
        const float h = dot(d, oc);
        const float c = norm_squared(oc) - md.radius * md.radius;
        const float disc = h * h - a * c;
        if (!(disc >= 0.0f)) continue;
        const float sq = sqrtf(disc);
        float t1 = (h - sq) / a, t2 = (h + sq) / a;
        if (t1 < 0.001f) t1 = 0.001f;
        if (t2 > best) t2 = best;
        if (!(t1 < t2)) continue;
        const float len = sqrtf(a);
        const float inside = (t2 - t1) * len;
        const float u = rng.range(0.0f, 1.0f);
        const float dist = (-1.0f / md.density) * logf(u);
        if (!(dist <= inside)) continue;
        best = t1 + dist / len;
        which = m;
        any = true;
    }
    t_event = best;
    return any;
}
// The share of light that gets from o to o + t_end * d without scattering: what a shadow ray of the next-event
// estimator is dimmed by (the path's own rays scatter instead, medium_event).
__device__ inline float medium_transmittance(const DevScene &sc, f3 o, f3 d, float t_end) {
    float tau = 0.0f;
    for (uint32_t m = 0; m < sc.n_media; m++) {
        const pts_medium md = sc.media[m];
        const f3 oc = mk3(md.center[0], md.center[1], md.center[2]) - o;
        const float a = norm_squared(d);
        const float h = dot(d, oc);
        const float c = norm_squared(oc) - md.radius * md.radius;
        const float disc = h * h - a * c;
        if (!(disc >= 0.0f)) continue;
        const float sq = sqrtf(disc);
        float t1 = (h - sq) / a, t2 = (h + sq) / a;
        if (t1 < 0.001f) t1 = 0.001f;
        if (t2 > t_end) t2 = t_end;
        if (!(t1 < t2)) continue;
        tau = tau + md.density * ((t2 - t1) * sqrtf(a));
    }
    return expf(-tau);
}
// Direction of the ray that leaves a scattering event: uniform on the sphere, two draws.
__device__ inline f3 medium_direction(Rng &rng) {
    const float u1 = rng.range(0.0f, 1.0f), u2 = rng.range(0.0f, 1.0f);
    const float z = 1.0f - 2.0f * u1;
    const float r = sqrtf(rmax(0.0f, 1.0f - z * z));
    const float phi = 2.0f * PTD_PI * u2;
    return mk3(r * cosf(phi), r * sinf(phi), z);
}

// ---------------------------------------------------------------- lights (src/objects.rs:235-278)
// LightBuffer::pdf: the sum over every light of its pdf for the ray, divided by the number of lights
// (src/objects.rs:259-278).  A light's pdf is the primitive's `hit` on a copy of the ray, then its
// formula; a light the ray misses adds 0, which leaves the f32 sum as it is.  So the sum over the
// lights that are hit, in the reference's order (triangles, spheres, quads), has the same bits as the
// sum over all of them.  Each light has a box in sc.light_boxes, wide enough that a ray which passes
// outside it is rejected by the exact test as well; the first loop tests up to 32 boxes with the same
// control flow in every lane and notes the survivors in a bit mask, the second runs the exact tests of
// those, lowest index first.  With many lights of a kind, groups of 32 consecutive ones have a box of their
// own (sc.light_group_boxes) and a lane opens only the groups its ray reaches.
//
// Triangle::pdf of the reference is 1 / area, a density over the triangle rather than over
// directions (spheres and quads report a solid-angle density).  TRIS_BY_SOLID_ANGLE converts it,
// distance^2 / (cosine * area) as Quad::pdf does; only the next-event estimator, which is not
// part of the reference and needs the density Triangle::sample really has, asks for that.
// Where a ray meets a light of the LightBuffer: distance, surface parameters, the light's index in the scene-wide
// array of its primitive type, and that type.
struct LightHit {
    float t, u, v;
    uint32_t prim, type;
};
template <bool TRIS_BY_SOLID_ANGLE>
__device__ inline float light_term(const DevScene &sc, uint32_t i, f3 o, f3 d, LightHit *where = nullptr) {
    const DevSet &L = sc.lights;
    const uint32_t nt = L.tris.prim_count, ns = L.spheres.prim_count, nq = L.quads.prim_count;
    TravCounters tc{0, 0, 0, 0};
    TravRay r = make_trav_ray(o, d, 0.001f, INFINITY);
    HitState h;
    h.code = PTD_HIT_MISS;
    h.u = h.v = 0.0f;
    h.prim = 0u;
    float term = 0.0f;
    if (i >= nt + ns + nq) return 0.0f;  // an instance in the light list: Hittable::pdf default (src/objects.rs:97-99)
    if (i < nt) {  // Triangle::pdf, tri_mesh.rs:252-261
        const uint32_t g = L.tris.prim_first + i;
        hit_prim<PT_PRIM_TRI>(sc, g, r, h, 0u, tc);
        if (h.code != PTD_HIT_MISS) {
            const float4 a = sc.tri_hot[3 * g], b = sc.tri_hot[3 * g + 1], c = sc.tri_hot[3 * g + 2];
            const f3 n = cross(mk3(a.w, b.x, b.y), mk3(b.z, b.w, c.x));
            if constexpr (TRIS_BY_SOLID_ANGLE) {
                const float nn = norm(n);
                const float distance_squared = r.t_max * r.t_max * norm_squared(d);
                const float cosine = fabsf(dot(d, n)) / (norm(d) * nn);
                term = distance_squared / (cosine * (0.5f * nn));
            } else {
                term = 2.0f / norm(n);
            }
        }
    } else if (i < nt + ns) {  // Sphere::pdf, sphere.rs:109-120
        const uint32_t g = L.spheres.prim_first + (i - nt);
        hit_prim<PT_PRIM_SPHERE>(sc, g, r, h, 0u, tc);
        if (h.code != PTD_HIT_MISS) {
            const float4 s = sc.spheres[g];
            float cos_theta_max = sqrtf(1.0f - s.w * s.w / norm_squared(mk3(s.x, s.y, s.z) - o));
            float solid_angle = 2.0f * PTD_PI * (1.0f - cos_theta_max);
            term = 1.0f / solid_angle;
        }
    } else {  // Quad::pdf, quad_mesh.rs:165-175
        const uint32_t g = L.quads.prim_first + (i - nt - ns);
        hit_prim<PT_PRIM_QUAD>(sc, g, r, h, 0u, tc);
        if (h.code != PTD_HIT_MISS) {
            const float4 q2 = sc.quads[4 * (size_t)g + 2];
            f3 qn = mk3(q2.y, q2.z, q2.w);
            // rec.normal() is the face-forwarded normal; |dot| makes the sign irrelevant
            float distance_squared = r.t_max * r.t_max * norm_squared(d);
            float cosine = fabsf(dot(d, qn)) / norm(d);
            term = distance_squared / (cosine * sc.quad_area[g]);
        }
    }
    if (where) {
        where->t = r.t_max;
        where->u = h.u;
        where->v = h.v;
        where->prim = h.prim;
        where->type = h.code & 3u;
    }
    return term;
}
// |a - b| within a ten-thousandth of b (of 1 below that): the same surface point along one ray
__device__ __forceinline__ bool same_distance(float a, float b) { return fabsf(a - b) <= 1.0e-4f * rmax(1.0f, fabsf(b)); }
typedef float light_box_t __attribute__((ext_vector_type(8)));  // lower corner, unused, upper corner, unused
#ifndef PT_LIGHTS_FLAT_MAX
#define PT_LIGHTS_FLAT_MAX 256u
#endif
constexpr uint32_t LIGHTS_FLAT_MAX = PT_LIGHTS_FLAT_MAX;  // more lights of one kind than this: boxes of groups of 32 are tested first
// AT_DISTANCE: only the lights that the ray meets at distance t_at count (next-event estimation: the density with
// which light sampling yields the very point a BSDF-sampled ray has reached, i.e. the emitter it hit, as a light).
template <bool TRIS_BY_SOLID_ANGLE = false, bool AT_DISTANCE = false>
__device__ inline float light_pdf(const DevScene &sc, f3 o, f3 d, float t_at = 0.0f) {
    const DevSet &L = sc.lights;
    // The boxes are read-only for as long as the scene lives and every lane reads the same one: through
    // the constant address space they come in over the scalar unit, one load per box for the wave.
    const __attribute__((address_space(4))) light_box_t *boxes = (const __attribute__((address_space(4))) light_box_t *)sc.light_boxes;
    const float o_max = rmax(fabsf(o.x), rmax(fabsf(o.y), fabsf(o.z)));
    // a component of the direction that is zero (or next to it) counts as +-1e-25: the ray then leaves a slab
    // it starts in only far beyond anything else, and no product below is 0 * inf
    auto inverse = [](float x) { return fabsf(x) > 1.0e-25f ? 1.0f / x : copysignf(1.0e25f, x); };
    const f3 inv = mk3(inverse(d.x), inverse(d.y), inverse(d.z));
    // Only rays of ordinary size are tested against the boxes.  Sphere::hit divides by |d|^2: when that is
    // zero, or anything is NaN, its root is NaN, which passes the range check, so such a ray "hits" every
    // sphere wherever it is; and a huge origin or direction would overflow the products below.
    const float dd = norm_squared(d);
    const bool ordinary = dd > 1.0e-30f && dd < 1.0e30f && o_max < 1.0e10f;
    float pdf = 0.0f;
    uint32_t first = 0, first_group = 0;
#pragma nounroll
    for (int kind = 0; kind < 3; kind++) {  // triangles, spheres, quads; instances: Hittable::pdf default, 0 (:97-99)
        const uint32_t n = kind == 0 ? L.tris.prim_count : (kind == 1 ? L.spheres.prim_count : L.quads.prim_count);
        const uint32_t n_groups = (n + 31u) / 32u;
        const float pad = (kind == 1 ? sc.light_pad_sphere : sc.light_pad_flat) * o_max;
        // (corner - origin) * inv as one fused multiply-add each; the rounding this changes is far inside the padding
        const f3 lo_c = mk3(-((o.x + pad) * inv.x), -((o.y + pad) * inv.y), -((o.z + pad) * inv.z));
        const f3 hi_c = mk3(-((o.x - pad) * inv.x), -((o.y - pad) * inv.y), -((o.z - pad) * inv.z));
        auto reaches = [&](const light_box_t b) {
            const float ax = __builtin_fmaf(b[0], inv.x, lo_c.x), bx = __builtin_fmaf(b[4], inv.x, hi_c.x);
            const float ay = __builtin_fmaf(b[1], inv.y, lo_c.y), by = __builtin_fmaf(b[5], inv.y, hi_c.y);
            const float az = __builtin_fmaf(b[2], inv.z, lo_c.z), bz = __builtin_fmaf(b[6], inv.z, hi_c.z);
            const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
            const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            return !(t_in > t_out);
        };
        auto all_of = [](uint32_t count) { return count == 32u ? 0xFFFFFFFFu : (1u << count) - 1u; };
        auto exact = [&](uint32_t base, uint32_t maybe) {  // the candidates of one group, lowest index first
            while (maybe != 0) {
                const uint32_t j = (uint32_t)__ffs((int)maybe) - 1u;
                maybe &= maybe - 1u;
                if constexpr (AT_DISTANCE) {
                    LightHit lh;
                    const float term = light_term<TRIS_BY_SOLID_ANGLE>(sc, base + j, o, d, &lh);
                    if (term != 0.0f && same_distance(lh.t, t_at)) pdf += term;
                } else {
                    pdf += light_term<TRIS_BY_SOLID_ANGLE>(sc, base + j, o, d);
                }
            }
        };
        if (n <= LIGHTS_FLAT_MAX) {
            // few lights: every lane looks at every box, the same one at the same time
            for (uint32_t g = 0; g < n_groups; g++) {
                const uint32_t base = first + 32u * g;
                const uint32_t count = first + n - base < 32u ? first + n - base : 32u;
                uint32_t maybe = 0;
#pragma unroll 8
                for (uint32_t j = 0; j < count; j++)
                    if (reaches(boxes[base + j])) maybe |= 1u << j;
                if (!ordinary) maybe = all_of(count);
                exact(base, maybe);
            }
        } else {
            // many lights: the boxes of whole groups of 32 first, the same way; then each lane looks into the
            // groups its own ray reaches, lowest first (the loads are no longer the same in every lane)
            const __attribute__((address_space(4))) light_box_t *group_boxes = (const __attribute__((address_space(4))) light_box_t *)sc.light_group_boxes;
            for (uint32_t g0 = 0; g0 < n_groups; g0 += 32u) {
                const uint32_t n_here = n_groups - g0 < 32u ? n_groups - g0 : 32u;
                uint32_t groups = 0;
#pragma unroll 8
                for (uint32_t j = 0; j < n_here; j++)
                    if (reaches(group_boxes[first_group + g0 + j])) groups |= 1u << j;
                if (!ordinary) groups = all_of(n_here);
                while (groups != 0) {
                    const uint32_t g = g0 + (uint32_t)__ffs((int)groups) - 1u;
                    groups &= groups - 1u;
                    const uint32_t base = first + 32u * g;
                    const uint32_t count = first + n - base < 32u ? first + n - base : 32u;
                    uint32_t maybe = 0;
#pragma nounroll
                    for (uint32_t j = 0; j < count; j++)
                        if (reaches(((const light_box_t *)sc.light_boxes)[base + j])) maybe |= 1u << j;
                    if (!ordinary) maybe = all_of(count);
                    exact(base, maybe);
                }
            }
        }
        first += n;
        first_group += n_groups;
    }
    return pdf / (float)sc.n_light_total;
}

// What light_pdf contributes when its weight in the mixture is zero (BSDF_SAMPLE_PROBABILITY = 1,
// src/camera.rs:317, :339-340): `0 * light_pdf` is 0 unless light_pdf is NaN or infinite.  Triangle and
// quad terms are finite whenever the primitive is hit at all (a hit needs a non-degenerate triangle,
// and a quad's |denominator| >= 1e-6).  A sphere's term, 1 / (2 pi (1 - sqrt(1 - r^2 / |c - o|^2))), is
// NaN when the origin lies inside the sphere and infinite when the sphere is so small or far that the
// root rounds to 1; it counts only if the ray hits that sphere.  So this returns 0, or NaN in those
// cases, from the sphere lights' own terms: the intersection test per light and vertex of the full sum
// is not needed for a product with zero.  Without any light the reference's average is 0 / 0: every
// vertex with a non-specular material then ends its path with nothing (src/camera.rs:355-357).
__device__ inline float light_pdf_times_zero(const DevScene &sc, f3 o, f3 d) {
    const DevSet &L = sc.lights;
    // an empty LightBuffer: the sum, 0, divided by the number of lights, 0 (src/objects.rs:260, :277)
    float r = sc.n_light_total == 0u ? NAN : 0.0f;
    TravCounters tc{0, 0, 0, 0};
    for (uint32_t i = 0; i < L.spheres.prim_count; i++) {
        const uint32_t g = L.spheres.prim_first + i;
        const float4 s = sc.spheres[g];
        const float cos_theta_max = sqrtf(1.0f - s.w * s.w / norm_squared(mk3(s.x, s.y, s.z) - o));
        const float term = 1.0f / (2.0f * PTD_PI * (1.0f - cos_theta_max));
        if (!(fabsf(term) <= 3.0e38f)) {
            TravRay ray = make_trav_ray(o, d, 0.001f, INFINITY);
            HitState h;
            h.code = PTD_HIT_MISS;
            hit_prim<PT_PRIM_SPHERE>(sc, g, ray, h, 0u, tc);
            if (h.code != PTD_HIT_MISS) r = NAN;
        }
    }
    return r;
}

// LightBuffer::sample: uniform pick, then the primitive's sampler; returns the unnormalised
// vector from `origin` towards the sampled point.
// Sphere::sample of the reference puts its cone around the basis' third axis while
// OrthonormalBasis::new puts the direction it is given on the second, so its samples point
// sideways and do not have the density Sphere::pdf reports.  SPHERES_TOWARDS swaps the two
// components; only the next-event estimator, which is not part of the reference, asks for that.
// *picked (may be null): the index of the light that was drawn.
template <bool SPHERES_TOWARDS = false>
__device__ inline f3 light_sample(const DevScene &sc, Rng &rng, f3 origin, uint32_t *picked = nullptr) {
    const DevSet &L = sc.lights;
    uint32_t i = rng.below(sc.n_light_total);
    const uint32_t nt = L.tris.prim_count, ns = L.spheres.prim_count, nq = L.quads.prim_count;
    if (picked) *picked = i;
    if (i < nt) {  // Triangle::sample, tri_mesh.rs:263-277
        const uint32_t g = L.tris.prim_first + i;
        const float4 a = sc.tri_hot[3 * g], b = sc.tri_hot[3 * g + 1], c = sc.tri_hot[3 * g + 2];
        f3 v0 = mk3(a.x, a.y, a.z), u = mk3(a.w, b.x, b.y), v = mk3(b.z, b.w, c.x);
        float s = rng.range(0.0f, 1.0f), t = rng.range(0.0f, 1.0f);
        f3 offset = (s + t <= 1.0f) ? (s * u + t * v) : ((1.0f - s) * u + (1.0f - t) * v);
        return (v0 - origin) + offset;
    }
    if (i < nt + ns) {  // Sphere::sample, sphere.rs:38-53, :122-128
        const float4 sp = sc.spheres[L.spheres.prim_first + (i - nt)];
        f3 direction = mk3(sp.x, sp.y, sp.z) - origin;
        float distance_squared = norm_squared(direction);
        Onb uvw = make_onb(direction);
        float r1 = rng.range(0.0f, 1.0f), r2 = rng.range(0.0f, 1.0f);
        float z = 1.0f + r2 * (sqrtf(1.0f - sp.w * sp.w / distance_squared) - 1.0f);
        float phi = 2.0f * PTD_PI * r1;
        float x = cosf(phi) * sqrtf(1.0f - z * z);
        float y = sinf(phi) * sqrtf(1.0f - z * z);
        const f3 c = mk3(x, y, z) * sp.w;
        return SPHERES_TOWARDS ? to_world(uvw, mk3(c.x, c.z, c.y)) : to_world(uvw, c);
    }
    if (i < nt + ns + nq) {  // Quad::sample, quad_mesh.rs:176-179
        const uint32_t g = L.quads.prim_first + (i - nt - ns);
        const float4 q0 = sc.quads[4 * (size_t)g], q1 = sc.quads[4 * (size_t)g + 1], q2 = sc.quads[4 * (size_t)g + 2];
        f3 qo = mk3(q0.x, q0.y, q0.z), qu = mk3(q0.w, q1.x, q1.y), qv = mk3(q1.z, q1.w, q2.x);
        float a = rng.range(0.0f, 1.0f);
        float b = rng.range(0.0f, 1.0f);
        f3 p = qo + qu * a + qv * b;
        return p - origin;
    }
    return mk3(0.0f, 0.0f, 0.0f);  // Hittable::sample default for instances (:100-102)
}

}  // namespace ptd
