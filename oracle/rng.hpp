// TEST INFRASTRUCTURE — CPU oracle (see na.hpp).
//
// Random numbers.  The reference draws from rand 0.8.5's ThreadRng
// (src/camera.rs:201; Cargo.lock:2886-2898), which is seeded by the OS, so no
// render of the reference is reproducible and no bit-level parity with it exists.
// What is restated here is (a) HOW MANY values each call site consumes and how a
// raw integer becomes the value the call site sees, following rand 0.8.5's
// published algorithms, and (b) a counter-based generator in place of ChaCha12 so
// that a (seed, pixel, stratum) triple fixes the whole path on any machine:
//
//   key   = mix64(mix64(seed) ^ ((uint64)stratum << 32 | pixel))
//   draw i (i = 1, 2, ...) = mix64(key + i * 0x9E3779B97F4A7C15)        (SplitMix64)
//
//   gen_range(lo..hi) for f32 (UniformFloat::sample_single): one 32-bit value w,
//       f = (w >> 9) * 2^-23  in [0,1);  result = f * (hi - lo) + lo
//       (the retry branch for result >= hi cannot trigger for the ranges used).
//   gen_bool(p) (Bernoulli): p == 1.0 draws nothing and is true; otherwise one
//       64-bit value v, true iff v < (uint64)(p * 2^64).
//   gen_range(0..n) for usize (UniformInt::sample_single): 64-bit v, widening
//       multiply by n, accept when the low half <= zone, result = high half.
// The 32-bit value of a draw is the upper half of the 64-bit output.
#pragma once
#include <cstdint>

namespace oracle {

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t key;
    uint32_t counter;

    static Rng for_sample(uint64_t seed, uint32_t pixel, uint32_t stratum) {
        Rng r;
        r.key = mix64(mix64(seed) ^ (((uint64_t)stratum << 32) | (uint64_t)pixel));
        r.counter = 0;
        return r;
    }
    uint64_t next_u64() {
        counter += 1;
        return mix64(key + (uint64_t)counter * 0x9E3779B97F4A7C15ull);
    }
    uint32_t next_u32() { return (uint32_t)(next_u64() >> 32); }
    float unit() { return (float)(next_u32() >> 9) * (1.0f / 8388608.0f); }
    float range(float lo, float hi) {
        float scale = hi - lo;
        return unit() * scale + lo;
    }
    bool gen_bool(double p) {
        if (p == 1.0) return true;
        // Bernoulli::new: p_int = (p * 2^64) as u64
        uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
        return next_u64() < p_int;
    }
    uint64_t below(uint64_t n) {
        unsigned lz = (unsigned)__builtin_clzll(n);
        uint64_t zone = (n << lz) - 1;
        for (;;) {
            uint64_t v = next_u64();
            unsigned __int128 m = (unsigned __int128)v * (unsigned __int128)n;
            uint64_t lo = (uint64_t)m, hi = (uint64_t)(m >> 64);
            if (lo <= zone) return hi;
        }
    }
};

}  // namespace oracle
