// Issue rate of wave64 vector instructions on one SIMD of gfx950, by instruction kind and by the number of
// waves the SIMD holds: independent instructions (eight accumulator registers in turn), cycles from s_memtime.
// Answers what "vector issue peak" a traversal kernel made of subtracts, multiplies, compares, selects and
// integer ops can be priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, BODY)                                                                          \
    __global__ void __launch_bounds__(1024) NAME(int iters, unsigned long long *cycles, float *sink, float seed) { \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        float b = seed * 0.5f, c = seed * 0.25f;                                                    \
        (void)c;                                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                       \
        for (int i = 0; i < iters; i++) {                                                           \
            BODY BODY BODY BODY                                                                     \
        }                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                       \
        if (threadIdx.x % 64 == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;  \
        float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                            \
        if (s == 12345.678f) sink[0] = s;                                                           \
    }

#define ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define MIN(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define IADD(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define LSH(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a##k));
#define MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a##k) : "v"(b));
#define CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : );
#define CNDS(k) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(b) : );
#define SUB(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MAXF(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define MIN3(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define XOR(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define BFE(k) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a##k));
#define LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a##k) : "v"(b));
#define ADD3(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
#define DIVFIX(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a##k) : "v"(b) : "vcc");
#define CVT(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##k));
#define CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##k), "v"(b) : "vcc");
#define CMPS(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a##k), "v"(b) : "s20", "s21");
#define CMPCND(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(b) : "vcc");
#define SUBMUL(k) asm volatile("v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %0, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define MADU(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##k) : "v"(b), "v"(c));
#define MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##k) : "v"(b));
#define RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
#define SALU(k) asm volatile("s_add_u32 s2%c0, s2%c0, 1" : : "n"(k) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
#define SBAL(k) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");
#define MIXVS(k) asm volatile("v_add_f32 %0, %0, %1\n s_and_b64 s[20:21], s[20:21], exec" : "+v"(a##k) : "v"(b) : "s20", "s21", "scc");

KERNEL(k_add, REP8(ADD))
KERNEL(k_mul, REP8(MUL))
KERNEL(k_fma, REP8(FMA))
KERNEL(k_min, REP8(MIN))
KERNEL(k_max3, REP8(MAX3))
KERNEL(k_iadd, REP8(IADD))
KERNEL(k_and, REP8(AND))
KERNEL(k_lsh, REP8(LSH))
KERNEL(k_mov, REP8(MOV))
KERNEL(k_cnd, REP8(CND))
KERNEL(k_cnds, REP8(CNDS))
KERNEL(k_sub, REP8(SUB))
KERNEL(k_maxf, REP8(MAXF))
KERNEL(k_min3, REP8(MIN3))
KERNEL(k_xor, REP8(XOR))
KERNEL(k_bfe, REP8(BFE))
KERNEL(k_lshladd, REP8(LSHLADD))
KERNEL(k_add3, REP8(ADD3))
KERNEL(k_sqrt, REP8(SQRT))
KERNEL(k_divfix, REP8(DIVFIX))
KERNEL(k_divscale, REP8(DIVSCALE))
KERNEL(k_cvt, REP8(CVT))
KERNEL(k_cmp, REP8(CMP))
KERNEL(k_cmps, REP8(CMPS))
KERNEL(k_cmpcnd, REP8(CMPCND))
KERNEL(k_submul, REP8(SUBMUL))
KERNEL(k_madu, REP8(MADU))
KERNEL(k_mullo, REP8(MULLO))
KERNEL(k_rcp, REP8(RCP))
KERNEL(k_sbal, REP8(SBAL))
KERNEL(k_mixvs, REP8(MIXVS))

// packed: two registers per operand
__global__ void __launch_bounds__(1024) k_pkadd(int iters, unsigned long long *cycles, float *sink, float seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = a0 * 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define PK(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a##k) : "v"(b));
    for (int i = 0; i < iters; i++) { REP8(PK) REP8(PK) REP8(PK) REP8(PK) }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x == 12345.678f) sink[0] = s.y;
}

typedef void (*kern_t)(int, unsigned long long *, float *, float);
struct Row { const char *name; kern_t k; int per_body; };

int main() {
    unsigned long long *cyc;
    float *sink;
    hipMalloc(&cyc, 1 << 20);
    hipMalloc(&sink, 4);
    const Row rows[] = {{"v_add_f32", k_add, 8}, {"v_mul_f32", k_mul, 8}, {"v_fma_f32", k_fma, 8}, {"v_min_f32", k_min, 8},
                        {"v_max3_f32", k_max3, 8}, {"v_add_u32", k_iadd, 8}, {"v_and_b32", k_and, 8}, {"v_lshlrev_b32", k_lsh, 8},
                        {"v_mov_b32", k_mov, 8}, {"v_cndmask_b32 (vcc)", k_cnd, 8}, {"v_cndmask_b32 (sgpr pair)", k_cnds, 8}, {"v_sub_f32", k_sub, 8}, {"v_max_f32", k_maxf, 8},
                        {"v_min3_f32", k_min3, 8}, {"v_xor_b32", k_xor, 8}, {"v_bfe_u32", k_bfe, 8}, {"v_lshl_add_u32", k_lshladd, 8}, {"v_add3_u32", k_add3, 8},
                        {"v_sqrt_f32", k_sqrt, 8}, {"v_div_fixup_f32", k_divfix, 8}, {"v_div_scale_f32", k_divscale, 8}, {"v_cvt_f32_u32", k_cvt, 8}, {"v_cmp_lt_f32 -> vcc", k_cmp, 8},
                        {"v_cmp_lt_f32 -> sgpr pair", k_cmps, 8}, {"v_cmp + v_cndmask (dependent through vcc)", k_cmpcnd, 16},
                        {"v_sub + v_mul (dependent)", k_submul, 16}, {"v_mad_u32_u24", k_madu, 8}, {"v_mul_lo_u32", k_mullo, 8},
                        {"v_rcp_f32", k_rcp, 8}, {"v_pk_add_f32", k_pkadd, 8}, {"s_and_b64 (scalar)", k_sbal, 8},
                        {"v_add_f32 + s_and_b64 interleaved", k_mixvs, 16}};
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int dev_clock_khz = 0;
    hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0);
    printf("ns per wave64 instruction per SIMD, wall clock over the whole chip (1024 SIMDs; device reports %d kHz); in brackets cycles at 2.4 GHz\n", dev_clock_khz);
    printf("%-45s", "waves per SIMD:");
    for (int w : {1, 2, 4, 8}) printf("%16d", w);
    printf("\n");
    const int iters = 20000;
    for (const Row &r : rows) {
        printf("%-45s", r.name);
        for (int wps : {1, 2, 4, 8}) {
            // workgroups of 4 * min(wps, 4) waves; 256 or 512 of them, which the dispatcher spreads over the 256 CUs
            const int threads = 64 * 4 * (wps < 4 ? wps : 4), grid = 256 * (wps <= 4 ? 1 : wps / 4);
            r.k<<<grid, threads>>>(10, cyc, sink, 1.0f);
            hipEventRecord(a);
            r.k<<<grid, threads>>>(iters, cyc, sink, 1.0f);
            hipEventRecord(b);
            if (hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) { printf("  launch failed"); continue; }
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double per_simd = (double)iters * 4 * r.per_body * wps;
            const double ns = ms * 1e6 / per_simd;
            printf("  %6.3f (%5.2f)", ns, ns * 2.4);
        }
        printf("\n");
    }
    return 0;
}
