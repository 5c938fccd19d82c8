// Device-side vector arithmetic and random numbers for the gfx950 kernels.
//
// Every expression is written in the order the reference evaluates it (nalgebra
// 0.33: dot = (x0*y0 + x1*y1) + x2*y2, normalize = divide by sqrt(norm_squared),
// matrix*vector accumulated column by column), and the file is compiled with FMA
// contraction off and correctly rounded f32 divide/sqrt, because first-hit
// primitive ids are compared in bits with a CPU evaluation of the same formulas.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ f3 cmul(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float norm_squared(f3 a) { return dot(a, a); }
__device__ __forceinline__ float norm(f3 a) { return sqrtf(norm_squared(a)); }
__device__ __forceinline__ f3 normalize(f3 a) {
    float n = norm(a);
    return {a.x / n, a.y / n, a.z / n};
}
// Vector::lerp(self=a, rhs=b, t) = t*b + (1-t)*a
__device__ __forceinline__ f3 lerp3(f3 a, f3 b, float t) {
    float s = 1.0f - t;
    return {t * b.x + s * a.x, t * b.y + s * a.y, t * b.z + s * a.z};
}
// Rust f32::min/max ignore a NaN operand; f32::clamp keeps NaN.
__device__ __forceinline__ float rmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float rclamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ bool any_nan(f3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }

#define PTD_PI 3.14159265358979323846f
#define PTD_INV_PI (1.0f / PTD_PI)

// 3x4 affine matrix rows (row-major): r0 = m[0..3], r1 = m[4..7], r2 = m[8..11].
struct Affine {
    float m[12];
};
// Matrix4::transform_vector: ((col0*x) + col1*y) + col2*z
__device__ __forceinline__ f3 xf_vector(const Affine &a, f3 v) {
    return {(a.m[0] * v.x + a.m[1] * v.y) + a.m[2] * v.z, (a.m[4] * v.x + a.m[5] * v.y) + a.m[6] * v.z,
            (a.m[8] * v.x + a.m[9] * v.y) + a.m[10] * v.z};
}
// Matrix4::transform_point for an affine matrix: M3*p + translation (the homogeneous divisor is exactly 1)
__device__ __forceinline__ f3 xf_point(const Affine &a, f3 p) {
    f3 r = xf_vector(a, p);
    return {r.x + a.m[3], r.y + a.m[7], r.z + a.m[11]};
}

// OrthonormalBasis::new (y = normal): columns u, v, w.
struct Onb {
    f3 u, v, w;
};
__device__ __forceinline__ Onb make_onb(f3 n) {
    Onb o;
    o.v = normalize(n);
    f3 a = (fabsf(o.v.x) > 0.9f) ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    o.w = normalize(cross(o.v, a));
    o.u = cross(o.v, o.w);
    return o;
}
__device__ __forceinline__ f3 to_local(const Onb &o, f3 x) { return {dot(o.u, x), dot(o.v, x), dot(o.w, x)}; }
__device__ __forceinline__ f3 to_world(const Onb &o, f3 x) {
    f3 y = o.u * x.x;
    y = o.v * x.y + y;
    y = o.w * x.z + y;
    return y;
}

// ---- counter-based generator -------------------------------------------------
// key = mix64(mix64(seed) ^ (stratum << 32 | pixel)); draw i = mix64(key + i * GOLDEN), i = 1, 2, ...
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// The time of a path (moving spheres): a hash of its generator's key, none of its draws.
__device__ __forceinline__ float path_time(uint64_t key) { return (float)(uint32_t)(mix64(key ^ 0x6D6F74696F6E5F74ull) >> 40) * (1.0f / 16777216.0f); }

struct Rng {
    uint64_t key;
    uint32_t counter;
    __device__ __forceinline__ void init(uint64_t seed_mixed, uint32_t pixel, uint32_t stratum, uint32_t ctr) {
        key = mix64(seed_mixed ^ (((uint64_t)stratum << 32) | (uint64_t)pixel));
        counter = ctr;
    }
    __device__ __forceinline__ uint64_t next_u64() {
        counter += 1;
        return mix64(key + (uint64_t)counter * 0x9E3779B97F4A7C15ull);
    }
    // gen_range(0.0..1.0): 23 random mantissa bits
    __device__ __forceinline__ float unit() { return (float)((uint32_t)(next_u64() >> 32) >> 9) * (1.0f / 8388608.0f); }
    __device__ __forceinline__ float range(float lo, float hi) {
        float scale = hi - lo;
        return unit() * scale + lo;
    }
    // gen_bool(p): p == 1 consumes nothing
    __device__ __forceinline__ bool gen_bool(float p) {
        if (p == 1.0f) return true;
        double pd = (double)p;
        uint64_t p_int = (uint64_t)(pd * 18446744073709551616.0);
        return next_u64() < p_int;
    }
    // gen_range(0..n)
    __device__ __forceinline__ uint32_t below(uint32_t n) {
        uint64_t nn = (uint64_t)n;
        uint64_t zone = (nn << __clzll((long long)nn)) - 1;
        for (;;) {
            uint64_t v = next_u64();
            uint64_t lo = v * nn;
            uint64_t hi = __umul64hi(v, nn);
            if (lo <= zone) return (uint32_t)hi;
        }
    }
};

// Rust `as u32` / `as i64` from f32: saturating, NaN -> 0
__device__ __forceinline__ uint32_t f2u32_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
__device__ __forceinline__ long long f2i64_sat(float f) {
    if (f != f) return 0;
    if (f >= 9.2233720368547758e18f) return 0x7FFFFFFFFFFFFFFFll;
    if (f <= -9.2233720368547758e18f) return (long long)0x8000000000000000ull;
    return (long long)f;
}

}  // namespace ptd
