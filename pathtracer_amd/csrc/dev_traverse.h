// Closest-hit search on the device: slab test, the three intersectors, flat-BVH
// traversal and instance descent.
//
// Order of operations follows the reference so that the same primitives are tested
// in the same sequence with the same f32 arithmetic:
//   BBox::intersect                       src/bvh.rs:145-182
//   InnerPrimitiveBuffer::traverse        src/bvh.rs:725-797   (near child first by sign of
//                                         the split axis, far child pushed, leaf in order,
//                                         a candidate with t == t_max is accepted)
//   Triangle::hit                         src/objects/tri_mesh.rs:167-198
//   Sphere::hit                           src/objects/sphere.rs:57-80
//   Quad::hit                             src/objects/quad_mesh.rs:126-147
//   Instance::hit                         src/objects.rs:384-401
//   PrimitiveBuffer::hit                  src/objects.rs:191-197 (triangles, spheres, quads, instances)
//
// Each lane keeps its traversal stack in LDS (one bank column per lane, so pushes and
// pops never conflict); entries beyond LDS_LEVELS go to a global overflow array.
#pragma once
#include "dev_queue.h"

namespace ptd {

constexpr int BLOCK = 256;      // threads per workgroup of the traversal kernels
#ifndef PT_LDS_LEVELS
#define PT_LDS_LEVELS 15
#endif
constexpr int LDS_LEVELS = PT_LDS_LEVELS;  // stack entries per lane held in LDS (8 B each: 30 KB per workgroup, five workgroups per CU)

struct TravRay {
    f3 o, d, inv_d;
    float t_min, t_max;
    uint32_t neg;  // bit a: d[a] < 0, what `ray.direction[axis] < 0.0` (src/bvh.rs:757) asks at every node with two children to visit
};
__device__ __forceinline__ TravRay make_trav_ray(f3 o, f3 d, float t_min, float t_max) {
    TravRay r;
    r.o = o;
    r.d = d;
    r.inv_d = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);  // `1.0 / ray.direction[axis]`, same value at every node
    r.t_min = t_min;
    r.t_max = t_max;
    r.neg = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    return r;
}

struct HitState {
    float u, v;        // barycentrics (triangle) / alpha, beta (quad)
    uint32_t prim;     // global primitive index
    uint32_t code;     // type | (instance+1) << 2, PTD_HIT_MISS while nothing was hit
};

struct TravCounters {
    uint32_t nodes, tris, spheres, quads;
};

// Per-lane traversal stack of (node reference, entry distance) pairs.
//   - the newest entry stays in two registers (most pushes are popped right away);
//   - the next LDS_LEVELS entries live in LDS, declared at namespace scope so that every
//     access is a ds_read/ds_write (a pointer to LDS kept in a struct decays to a generic
//     pointer and turns into flat_load/flat_store), laid out [level][lane] so that a wave's
//     accesses fall on distinct banks;
//   - deeper entries go to a global overflow array [level][launch-wide lane id]; the host
//     sizes it from the depth of the uploaded BVHs, so it cannot be overrun.
__shared__ uint2 g_lane_stack[LDS_LEVELS * BLOCK];

// The overflow array is addressed through a pointer that says "global memory" in its type:
// a pop chooses between an LDS and a global load, and with a generic pointer on one side the
// compiler folds the two into one flat_load, which pays for both paths on every pop.
typedef __attribute__((address_space(1))) uint2 GlobalUint2;

// "No entry in the register slot" is a value of the slot itself, not a flag beside it: a per-lane flag lives in a
// scalar register pair, and every place where two control-flow paths meet merges it with three scalar instructions;
// the scalar unit is shared by the four SIMDs of a CU (one instruction per cycle per CU, measured), so those count.
constexpr uint32_t STACK_NO_TOP = 0xFFFFFFFFu;  // never pushed: node references and REF_* markers are all below it

struct LaneStack {
    GlobalUint2 *overflow;  // [levels][stride]
    uint32_t stride;  // lanes in the launch
    uint32_t gtid;
    uint32_t top_ref;  // STACK_NO_TOP: the register slot is empty
    float top_e;
    int sp;  // entries below the register slot
    __device__ __forceinline__ void init(uint2 *ovf, uint32_t lanes) {
        overflow = (GlobalUint2 *)ovf;
        stride = lanes;
        gtid = blockIdx.x * BLOCK + threadIdx.x;
        sp = 0;
        top_ref = STACK_NO_TOP;
        top_e = 0.0f;
    }
    __device__ __forceinline__ bool has_top() const { return top_ref != STACK_NO_TOP; }
    __device__ __forceinline__ void clear() {
        sp = 0;
        top_ref = STACK_NO_TOP;
    }
    // The lanes that are here together nearly always all have their level in LDS: one scalar compare and branch
    // decides that for the wave, and only a wave with a lane beyond LDS_LEVELS chooses per lane.
    __device__ __forceinline__ void spill_top() {
        const uint2 v = make_uint2(top_ref, __float_as_uint(top_e));
        if (__ballot(sp >= LDS_LEVELS) == 0ull) {
            g_lane_stack[sp * BLOCK + threadIdx.x] = v;
        } else if (sp < LDS_LEVELS) {
            g_lane_stack[sp * BLOCK + threadIdx.x] = v;
        } else {
            GlobalUint2 *p = overflow + ((size_t)(sp - LDS_LEVELS) * stride + gtid);
            p->x = v.x;
            p->y = v.y;
        }
        sp++;
    }
    __device__ __forceinline__ void push(uint32_t ref, float e) {
        if (has_top()) spill_top();
        top_ref = ref;
        top_e = e;
    }
    // Entries below `floor` belong to an enclosing traversal.
    __device__ __forceinline__ bool pop(int floor, uint32_t &ref, float &e) {
        if (has_top()) {
            ref = top_ref;
            e = top_e;
            top_ref = STACK_NO_TOP;
            return true;
        }
        if (sp == floor) return false;
        sp--;
        uint2 v;
        if (__ballot(sp >= LDS_LEVELS) == 0ull) {
            v = g_lane_stack[sp * BLOCK + threadIdx.x];
        } else if (sp < LDS_LEVELS) {
            v = g_lane_stack[sp * BLOCK + threadIdx.x];
        } else {
            const GlobalUint2 *p = overflow + ((size_t)(sp - LDS_LEVELS) * stride + gtid);
            v = make_uint2(p->x, p->y);
        }
        ref = v.x;
        e = __uint_as_float(v.y);
        return true;
    }
    // push and pop without a branch per lane, for the persistent-wave kernels, where the lanes that are here together
    // have each their own reason to be.  While every lane's level is in LDS (one scalar compare for the wave):
    // a push writes the register slot to the level above the stack whether it holds an entry or not, and only the
    // count says which; a pop reads the level below whether it will be needed or not.  Same stack contents.
    __device__ __forceinline__ void push_flat(uint32_t ref, float e) {
        if (__ballot(sp >= LDS_LEVELS) == 0ull) {
            g_lane_stack[sp * BLOCK + threadIdx.x] = make_uint2(top_ref, __float_as_uint(top_e));
            sp += has_top() ? 1 : 0;
        } else if (has_top()) {
            spill_top();
        }
        top_ref = ref;
        top_e = e;
    }
    __device__ __forceinline__ bool pop_flat(uint32_t &ref, float &e) {
        if (__ballot(sp > LDS_LEVELS) == 0ull) {
            const uint2 v = g_lane_stack[(sp > 0 ? sp - 1 : 0) * BLOCK + threadIdx.x];
            const bool top = has_top(), more = sp > 0;
            ref = top ? top_ref : v.x;
            e = top ? top_e : __uint_as_float(v.y);
            sp -= (!top && more) ? 1 : 0;
            top_ref = STACK_NO_TOP;
            return top || more;
        }
        return pop(0, ref, e);
    }
    // Start of a nested traversal: everything pushed so far must sit below its floor.
    __device__ __forceinline__ int enter() {
        if (has_top()) {
            spill_top();
            top_ref = STACK_NO_TOP;
        }
        return sp;
    }
};

// Per-axis update of BBox::intersect; both arms of `if t0 < t1` assign the smaller of
// (t0,t1) to the entry side and the larger to the exit side, and a NaN takes the else arm.
__device__ __forceinline__ void slab_axis(float lo, float hi, float o, float inv, float &t_min, float &t_max) {
    float t0 = (lo - o) * inv;
    float t1 = (hi - o) * inv;
    bool lt = t0 < t1;
    float tn = lt ? t0 : t1;
    float tf = lt ? t1 : t0;
    if (tn > t_min) t_min = tn;
    if (tf < t_max) t_max = tf;
}
// The reference leaves the loop at the first axis with t_max < t_min; t_min only grows and
// t_max only shrinks, so testing once after the third axis gives the same answer.
__device__ __forceinline__ bool slab_test(float4 n0, float4 n1, const TravRay &r) {
    float t_min = r.t_min, t_max = r.t_max;
    slab_axis(n0.x, n0.w, r.o.x, r.inv_d.x, t_min, t_max);
    slab_axis(n0.y, n1.x, r.o.y, r.inv_d.y, t_min, t_max);
    slab_axis(n0.z, n1.y, r.o.z, r.inv_d.z, t_min, t_max);
    return !(t_max < t_min);
}
// The same test on a box given as six scalars; also returns the entry distance
// E = max(ray.t_min, near planes), which does not depend on ray.t_max.  A box that passed
// with some t_max passes with a smaller t_max' exactly when !(t_max' < E): the exit side is
// min(t_max', X) with X = min(far planes), and the earlier pass already showed X >= E.
__device__ __forceinline__ bool slab_test_entry(float lx, float ly, float lz, float hx, float hy, float hz, const TravRay &r, float &entry) {
    float t_min = r.t_min, t_max = r.t_max;
    slab_axis(lx, hx, r.o.x, r.inv_d.x, t_min, t_max);
    slab_axis(ly, hy, r.o.y, r.inv_d.y, t_min, t_max);
    slab_axis(lz, hz, r.o.z, r.inv_d.z, t_min, t_max);
    entry = t_min;
    return !(t_max < t_min);
}

// slab_test_entry for rays on which no product in it can be a NaN: the origin is finite and
// each 1/d is finite and not zero (a NaN needs 0 * inf or inf - inf), the box being finite
// too.  Without NaNs each select of the exact form is a min or a max: (t0 < t1 ? t0 : t1)
// is min(t0, t1), `if (tn > t_min) t_min = tn` is max(t_min, tn), and so on; a zero of the
// other sign may come out, which no comparison downstream can tell apart.
__device__ __forceinline__ bool ray_is_plain(const TravRay &r) {
    const float big = 3.0e38f;  // finite and nonzero: 0 < |x| <= big is false for 0, inf and NaN
    return fabsf(r.inv_d.x) <= big && fabsf(r.inv_d.y) <= big && fabsf(r.inv_d.z) <= big && r.inv_d.x != 0.0f && r.inv_d.y != 0.0f &&
           r.inv_d.z != 0.0f && fabsf(r.o.x) <= big && fabsf(r.o.y) <= big && fabsf(r.o.z) <= big;
}
// (The packed form below, applied to this one-box test with the coordinates paired as a 32-byte node keeps them, was
// measured slower in the wave-at-a-time kernel: 11 more registers for the rotated copies of origin and 1/d, 4.99 ms
// per launch against 4.51.)
__device__ __forceinline__ bool slab_test_entry_plain(float lx, float ly, float lz, float hx, float hy, float hz, const TravRay &r,
                                                      float &entry) {
    const float ax = (lx - r.o.x) * r.inv_d.x, bx = (hx - r.o.x) * r.inv_d.x;
    const float ay = (ly - r.o.y) * r.inv_d.y, by = (hy - r.o.y) * r.inv_d.y;
    const float az = (lz - r.o.z) * r.inv_d.z, bz = (hz - r.o.z) * r.inv_d.z;
    const float t_min = fmaxf(fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)), r.t_min);
    const float t_max = fminf(fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz)), r.t_max);
    entry = t_min;
    return !(t_max < t_min);
}
// The same with the near and far plane of each axis chosen by the sign of the ray's direction instead of by
// min / max: lo <= hi, and subtraction and multiplication round monotonically, so (lo - o) * inv <= (hi - o) * inv
// exactly when inv > 0, and the smaller of the two is the one the sign names (equal values: either).  nx, ny, nz:
// d < 0 on that axis, which for a plain ray is inv < 0.  With the three conditions as wave masks in scalar registers
// the wave-at-a-time kernel's launches are 2 % shorter than with min / max (measured; in isolation a select on a mask
// in scalar registers issues at the rate of v_min_f32, tools/micro/valu_rate.hip, so the gain is in what the compiler
// makes of the rest); the same in the ray-drawing kernel, whose lanes change rays, was within noise and is not used.
__device__ __forceinline__ bool slab_test_entry_signed(float lx, float ly, float lz, float hx, float hy, float hz, const TravRay &r, bool nx, bool ny,
                                                       bool nz, float &entry) {
    const float ax = (lx - r.o.x) * r.inv_d.x, bx = (hx - r.o.x) * r.inv_d.x;
    const float ay = (ly - r.o.y) * r.inv_d.y, by = (hy - r.o.y) * r.inv_d.y;
    const float az = (lz - r.o.z) * r.inv_d.z, bz = (hz - r.o.z) * r.inv_d.z;
    const float t_min = fmaxf(fmaxf(fmaxf(nx ? bx : ax, ny ? by : ay), nz ? bz : az), r.t_min);
    const float t_max = fminf(fminf(fminf(nx ? ax : bx, ny ? ay : by), nz ? az : bz), r.t_max);
    entry = t_min;
    return !(t_max < t_min);
}
typedef float v2f __attribute__((ext_vector_type(2)));


// Both boxes of a 64-byte node for a plain ray (slab_test_entry_plain twice).  The node keeps its twelve
// coordinates as a = (lmin.x, lmin.y, lmax.x, lmax.y), b = (lmin.z, lmax.z, rmin.z, rmax.z), c = (rmin.x, rmin.y,
// rmax.x, rmax.y), so that the subtractions and multiplications go two at a time through the packed single-precision
// instructions (v_pk_add_f32, v_pk_mul_f32: one issue slot for two results, each rounded as the scalar form rounds it):
// twelve instructions instead of twenty-four in a kernel that is bound by the number of vector instructions it issues.
__device__ __forceinline__ void slab_test_two_plain(float4 a, float4 b, float4 c, const TravRay &r, bool &hit_l, float &entry_l, bool &hit_r, float &entry_r) {
    const v2f oxy = {r.o.x, r.o.y}, ozz = {r.o.z, r.o.z}, ixy = {r.inv_d.x, r.inv_d.y}, izz = {r.inv_d.z, r.inv_d.z};
    const v2f l0 = ((v2f){a.x, a.y} - oxy) * ixy, l1 = ((v2f){a.z, a.w} - oxy) * ixy, lz = ((v2f){b.x, b.y} - ozz) * izz;
    const v2f rz = ((v2f){b.z, b.w} - ozz) * izz, r0 = ((v2f){c.x, c.y} - oxy) * ixy, r1 = ((v2f){c.z, c.w} - oxy) * ixy;
    const float l_min = fmaxf(fmaxf(fmaxf(fminf(l0.x, l1.x), fminf(l0.y, l1.y)), fminf(lz.x, lz.y)), r.t_min);
    const float l_max = fminf(fminf(fminf(fmaxf(l0.x, l1.x), fmaxf(l0.y, l1.y)), fmaxf(lz.x, lz.y)), r.t_max);
    const float r_min = fmaxf(fmaxf(fmaxf(fminf(r0.x, r1.x), fminf(r0.y, r1.y)), fminf(rz.x, rz.y)), r.t_min);
    const float r_max = fminf(fminf(fminf(fmaxf(r0.x, r1.x), fmaxf(r0.y, r1.y)), fmaxf(rz.x, rz.y)), r.t_max);
    entry_l = l_min;
    entry_r = r_min;
    hit_l = !(l_max < l_min);
    hit_r = !(r_max < r_min);
}

template <int TYPE>
__device__ __forceinline__ void hit_prim(const DevScene &sc, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code,
                                         TravCounters &tc);

// Triangle::hit (src/objects/tri_mesh.rs:167-198) on the three float4 of tri_hot: v0, e1, e2.
__device__ __forceinline__ void hit_tri(float4 a, float4 b, float4 c, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code) {
    f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
    f3 e1_x_d = cross(e1, r.d);
    float triple = dot(e1_x_d, e2);
    if (triple == 0.0f) return;
    float inv_triple = 1.0f / triple;
    f3 s = r.o - v0;
    f3 e2_x_s = cross(e2, s);
    float u = dot(e2_x_s, r.d) * inv_triple;
    if (u < 0.0f || u > 1.0f) return;
    float v = dot(e1_x_d, s) * inv_triple;
    if (v < 0.0f || u + v > 1.0f) return;
    float t = dot(e2_x_s, e1) * inv_triple;
    if (!(r.t_min <= t && t <= r.t_max)) return;
    r.t_max = t;
    h.u = u;
    h.v = v;
    h.prim = g;
    h.code = PT_PRIM_TRI | inst_code;
}
// The same for the closest-hit kernels, without the early exits.  The reference leaves at the first condition that
// fails; a wave leaves only when all its lanes do, which a handful of rays at one triangle each practically never
// do together, so there the exits are only branches (three scalar instructions each).  Everything is computed and
// the conditions, each as the reference states it (a NaN passes `u < 0 || u > 1` and fails the range test of t),
// decide together whether the hit is taken.  (The shading kernels keep the form above for their light tests: this
// one costs them 3 to 6 registers and with them a wave per SIMD.)
__device__ __forceinline__ void hit_tri_flat(float4 a, float4 b, float4 c, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code) {
    f3 v0 = mk3(a.x, a.y, a.z), e1 = mk3(a.w, b.x, b.y), e2 = mk3(b.z, b.w, c.x);
    f3 e1_x_d = cross(e1, r.d);
    float triple = dot(e1_x_d, e2);
    float inv_triple = 1.0f / triple;
    f3 s = r.o - v0;
    f3 e2_x_s = cross(e2, s);
    float u = dot(e2_x_s, r.d) * inv_triple;
    float v = dot(e1_x_d, s) * inv_triple;
    float t = dot(e2_x_s, e1) * inv_triple;
    const bool take = (triple != 0.0f) & !((u < 0.0f) | (u > 1.0f)) & !((v < 0.0f) | (u + v > 1.0f)) & ((r.t_min <= t) & (t <= r.t_max));
    r.t_max = take ? t : r.t_max;
    h.u = take ? u : h.u;
    h.v = take ? v : h.v;
    h.prim = take ? g : h.prim;
    h.code = take ? (PT_PRIM_TRI | inst_code) : h.code;
}
__device__ __forceinline__ void hit_tri_flat(const DevScene &sc, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code, TravCounters &tc) {
    tc.tris++;
    hit_tri_flat(sc.tri_hot[3 * g], sc.tri_hot[3 * g + 1], sc.tri_hot[3 * g + 2], g, r, h, inst_code);
}
template <>
__device__ __forceinline__ void hit_prim<PT_PRIM_TRI>(const DevScene &sc, uint32_t g, TravRay &r, HitState &h,
                                                      uint32_t inst_code, TravCounters &tc) {
    tc.tris++;
    hit_tri(sc.tri_hot[3 * g], sc.tri_hot[3 * g + 1], sc.tri_hot[3 * g + 2], g, r, h, inst_code);
}

// Sphere::hit (src/objects/sphere.rs:56-99) for sphere g with its centre and radius in s.
__device__ __forceinline__ void hit_sphere(float4 s, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code);
template <>
__device__ __forceinline__ void hit_prim<PT_PRIM_SPHERE>(const DevScene &sc, uint32_t g, TravRay &r, HitState &h,
                                                         uint32_t inst_code, TravCounters &tc) {
    tc.spheres++;
    hit_sphere(sc.spheres[g], g, r, h, inst_code);
}
__device__ __forceinline__ void hit_sphere(float4 s, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code) {
    f3 oc = mk3(s.x, s.y, s.z) - r.o;
    float a = norm_squared(r.d);
    float hh = dot(r.d, oc);
    float c = norm_squared(oc) - s.w * s.w;
    float discriminant = hh * hh - a * c;
    if (discriminant < 0.0f) return;
    float sqrt_disc = sqrtf(discriminant);
    float root = (hh - sqrt_disc) / a;
    if (root < r.t_min || r.t_max < root) {
        root = (hh + sqrt_disc) / a;
        if (root < r.t_min || r.t_max < root) return;
    }
    r.t_max = root;
    h.u = 0.0f;
    h.v = 0.0f;
    h.prim = g;
    h.code = PT_PRIM_SPHERE | inst_code;
}

// Quad::hit (src/objects/quad_mesh.rs:126-147) on the four float4 of `quads`.
__device__ __forceinline__ void hit_quad(float4 q0, float4 q1, float4 q2, float4 q3, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code) {
    f3 origin = mk3(q0.x, q0.y, q0.z), qu = mk3(q0.w, q1.x, q1.y), qv = mk3(q1.z, q1.w, q2.x);
    f3 normal = mk3(q2.y, q2.z, q2.w), w = mk3(q3.x, q3.y, q3.z);
    float d = q3.w;
    float denom = dot(normal, r.d);
    if (fabsf(denom) < 1e-6f) return;
    float t = (d - dot(normal, r.o)) / denom;
    if (t < r.t_min || t > r.t_max) return;
    f3 intersection = r.o + t * r.d;
    f3 p = intersection - origin;
    float alpha = dot(w, cross(p, qv));
    float beta = dot(w, cross(qu, p));
    if (!(0.0f <= alpha && alpha <= 1.0f) || !(0.0f <= beta && beta <= 1.0f)) return;
    r.t_max = t;
    h.u = alpha;
    h.v = beta;
    h.prim = g;
    h.code = PT_PRIM_QUAD | inst_code;
}
// Without the early exits, for the closest-hit kernels (see hit_tri_flat): the same conditions, each as the reference
// states it, decide together.
__device__ __forceinline__ void hit_quad_flat(float4 q0, float4 q1, float4 q2, float4 q3, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code) {
    f3 origin = mk3(q0.x, q0.y, q0.z), qu = mk3(q0.w, q1.x, q1.y), qv = mk3(q1.z, q1.w, q2.x);
    f3 normal = mk3(q2.y, q2.z, q2.w), w = mk3(q3.x, q3.y, q3.z);
    float d = q3.w;
    float denom = dot(normal, r.d);
    float t = (d - dot(normal, r.o)) / denom;
    f3 intersection = r.o + t * r.d;
    f3 p = intersection - origin;
    float alpha = dot(w, cross(p, qv));
    float beta = dot(w, cross(qu, p));
    const bool take = !(fabsf(denom) < 1e-6f) & !((t < r.t_min) | (t > r.t_max)) & ((0.0f <= alpha) & (alpha <= 1.0f)) & ((0.0f <= beta) & (beta <= 1.0f));
    r.t_max = take ? t : r.t_max;
    h.u = take ? alpha : h.u;
    h.v = take ? beta : h.v;
    h.prim = take ? g : h.prim;
    h.code = take ? (PT_PRIM_QUAD | inst_code) : h.code;
}
__device__ __forceinline__ void hit_quad_flat(const DevScene &sc, uint32_t g, TravRay &r, HitState &h, uint32_t inst_code, TravCounters &tc) {
    tc.quads++;
    hit_quad_flat(sc.quads[4 * g], sc.quads[4 * g + 1], sc.quads[4 * g + 2], sc.quads[4 * g + 3], g, r, h, inst_code);
}
template <>
__device__ __forceinline__ void hit_prim<PT_PRIM_QUAD>(const DevScene &sc, uint32_t g, TravRay &r, HitState &h,
                                                       uint32_t inst_code, TravCounters &tc) {
    tc.quads++;
    hit_quad(sc.quads[4 * g], sc.quads[4 * g + 1], sc.quads[4 * g + 2], sc.quads[4 * g + 3], g, r, h, inst_code);
}

// One lane's search of a buffer whose tree was built on the device (64-byte two-box nodes only): the order of
// tests the ray-drawing search (dev_walk.h) makes for it: both boxes at the parent, the near child first, the far
// one stacked with its entry distance and dropped when popped if t_max has shrunk below it.
template <class LeafFn>
__device__ __noinline__ void traverse_wide(const DevScene &sc, const DevBuffer &buf, TravRay &r, LaneStack &st, TravCounters &tc, LeafFn &&leaf) {
        float e;
        tc.nodes++;
        if (!slab_test_entry(buf.rmin[0], buf.rmin[1], buf.rmin[2], buf.rmax[0], buf.rmax[1], buf.rmax[2], r, e)) return;
        const int floor = st.enter();
        const float4 *wide = sc.wide + 4 * (size_t)buf.wide_first;
        uint32_t cur = buf.root_ref;
        for (;;) {
            bool pop = true;
            if (cur & WIDE_LEAF) {  // WIDE_LEAF | count << 24 | offset
                const uint32_t first = buf.prim_first + (cur & 0xFFFFFFu), cnt = (cur >> 24) & 0x7Fu;
#pragma unroll 1
                for (uint32_t i = 0; i < cnt; i++) leaf(first + i);
            } else {
                const float4 a = wide[4 * (size_t)cur], b = wide[4 * (size_t)cur + 1], c = wide[4 * (size_t)cur + 2], d = wide[4 * (size_t)cur + 3];
                float el, er;
                tc.nodes += 2;
                const bool hl = slab_test_entry(a.x, a.y, b.x, a.z, a.w, b.y, r, el);
                const bool hr = slab_test_entry(c.x, c.y, b.z, c.z, c.w, b.w, r, er);
                const uint32_t lref = __float_as_uint(d.x), rref = __float_as_uint(d.y), axis = __float_as_uint(d.z);
                const float da = (axis == 0) ? r.d.x : ((axis == 1) ? r.d.y : r.d.z);
                const bool neg = da < 0.0f;
                const uint32_t near_ref = neg ? rref : lref, far_ref = neg ? lref : rref;
                const bool hn = neg ? hr : hl, hf = neg ? hl : hr;
                if (hn) {
                    if (hf) st.push(far_ref, neg ? el : er);
                    cur = near_ref;
                    pop = false;
                } else if (hf) {
                    cur = far_ref;
                    pop = false;
                }
            }
            if (pop) {
                bool got = false;
                while (st.pop(floor, cur, e)) {
                    if (!(r.t_max < e)) {
                        got = true;
                        break;
                    }
                }
                if (!got) break;
            }
        }
}

// ---------------------------------------------------------------------------------------
// The reference's loop verbatim over 32-byte nodes (InnerPrimitiveBuffer::traverse,
// src/bvh.rs:725-797): one AABB per visit, near child first by the sign of the direction on
// the split axis, the far child pushed untested and tested when popped.  Used when AABB /
// primitive tests are being counted, when a scene does not fit the 64-byte node layout, and
// as the yardstick the persistent-wave traversal (dev_walk.h) is checked against.
template <class LeafFn>
__device__ __forceinline__ void traverse(const DevScene &sc, const DevBuffer &buf, TravRay &r, LaneStack &st, TravCounters &tc,
                                         LeafFn &&leaf) {
    if (buf.prim_count == 0) return;
    if (buf.tree == 2u) {
        traverse_wide(sc, buf, r, st, tc, leaf);
        return;
    }
    if (buf.node_count == 0) {  // no BVH: src/bvh.rs:726-737
#pragma unroll 1
        for (uint32_t i = 0; i < buf.prim_count; i++) leaf(buf.prim_first + i);
        return;
    }
    const bool neg_x = r.d.x < 0.0f, neg_y = r.d.y < 0.0f, neg_z = r.d.z < 0.0f;
    const int floor = st.enter();
    const float4 *nodes = sc.nodes + 2 * (size_t)buf.node_first;
    uint32_t cur = 0;
    for (;;) {
        const float4 n0 = nodes[2 * cur], n1 = nodes[2 * cur + 1];
        tc.nodes++;
        bool descend = false;
        if (slab_test(n0, n1, r)) {
            const uint32_t a = __float_as_uint(n1.z), b = __float_as_uint(n1.w);
            if (!(b & PT_NODE_LEAF)) {
                const bool neg = (b == 0) ? neg_x : ((b == 1) ? neg_y : neg_z);
                if (neg) {
                    st.push(cur + 1, 0.0f);
                    cur = a;
                } else {
                    st.push(a, 0.0f);
                    cur = cur + 1;
                }
                descend = true;
            } else {
                const uint32_t cnt = b & 0xFFFFu;
#pragma unroll 1
                for (uint32_t i = 0; i < cnt; i++) leaf(buf.prim_first + a + i);
            }
        }
        if (!descend) {
            float e;
            if (!st.pop(floor, cur, e)) break;
        }
    }
}

// The triangle, sphere and quad buffers of one PrimitiveBuffer, in that order.
__device__ __forceinline__ void hit_set_prims(const DevScene &sc, const DevSet &set, TravRay &r, HitState &h,
                                              uint32_t inst_code, LaneStack &st, TravCounters &tc) {
    traverse(sc, set.tris, r, st, tc, [&](uint32_t g) { hit_prim<PT_PRIM_TRI>(sc, g, r, h, inst_code, tc); });
    traverse(sc, set.spheres, r, st, tc, [&](uint32_t g) { hit_prim<PT_PRIM_SPHERE>(sc, g, r, h, inst_code, tc); });
    traverse(sc, set.quads, r, st, tc, [&](uint32_t g) { hit_prim<PT_PRIM_QUAD>(sc, g, r, h, inst_code, tc); });
}

__device__ __forceinline__ Affine load_affine(const float4 *p) {
    Affine a;
    const float4 r0 = p[0], r1 = p[1], r2 = p[2];
    a.m[0] = r0.x; a.m[1] = r0.y; a.m[2] = r0.z; a.m[3] = r0.w;
    a.m[4] = r1.x; a.m[5] = r1.y; a.m[6] = r1.z; a.m[7] = r1.w;
    a.m[8] = r2.x; a.m[9] = r2.y; a.m[10] = r2.z; a.m[11] = r2.w;
    return a;
}

// An instance that sat inside another instance's set is searched from the top level (the host layer lifts it
// there), but keeps the steps of its nesting so that every number comes out as the reference's level-by-level
// Instance::hit computes it (src/objects.rs:385-397): its own record is the outermost instance, records
// chain_first .. chain_first + chain_count - 1 of the instance array (which belong to no set) are the ones further
// in, outer to inner.  `last` is the record's seventh float4: (child set, chain_first, chain_count, -).
__device__ __forceinline__ void chain_into(const DevScene &sc, float4 last, f3 &o, f3 &d) {
    const uint32_t n = __float_as_uint(last.z);
    const float4 *cp = sc.insts + 7 * (size_t)__float_as_uint(last.y);
    for (uint32_t k = 0; k < n; k++, cp += 7) {
        const Affine inv = load_affine(cp);
        o = xf_point(inv, o);
        d = xf_vector(inv, d);
    }
}
// ... and the way back for the hit point and the normal: the innermost `transform` first (:393-394 at each level)
__device__ __forceinline__ void chain_out_of(const DevScene &sc, float4 last, f3 &p, f3 &n) {
    const float4 *cp = sc.insts + 7 * (size_t)__float_as_uint(last.y);
    for (uint32_t k = __float_as_uint(last.z); k-- > 0u;) {
        const Affine fwd = load_affine(cp + 7 * (size_t)k + 3);
        p = xf_point(fwd, p);
        n = xf_vector(fwd, n);
    }
}

// Instance::hit: the ray is taken into the instance's frame without renormalising the
// direction, so t is shared between the two frames.
__device__ __forceinline__ void hit_instance(const DevScene &sc, uint32_t gi, TravRay &r, HitState &h, LaneStack &st,
                                             TravCounters &tc) {
    const float4 *ip = sc.insts + 7 * (size_t)gi;
    const Affine inv = load_affine(ip);
    const float4 last = ip[6];
    const uint32_t child = __float_as_uint(last.x);
    f3 lo = xf_point(inv, r.o), ld = xf_vector(inv, r.d);
    chain_into(sc, last, lo, ld);
    TravRay local = make_trav_ray(lo, ld, r.t_min, r.t_max);
    HitState hl;
    hl.code = PTD_HIT_MISS;
    hl.u = hl.v = 0.0f;
    hl.prim = 0;
    const DevSet &set = sc.sets[child];
    hit_set_prims(sc, set, local, hl, (gi - sc.world.insts.prim_first + 1u) << 2, st, tc);
    if (hl.code != PTD_HIT_MISS) {
        h = hl;
        r.t_max = local.t_max;
    }
}

// PrimitiveBuffer::hit on the world set, in two parts that run as two kernels (the second
// only for scenes that have instances): first the triangle, sphere and quad buffers, then
// the instance buffer, which the reference also visits last, with t_max as the first part
// left it.  On return r.t_max is the hit distance (or the incoming t_max on a miss) and h
// identifies the primitive.
__device__ __forceinline__ void hit_world_prims(const DevScene &sc, TravRay &r, HitState &h, LaneStack &st, TravCounters &tc) {
    h.code = PTD_HIT_MISS;
    h.u = h.v = 0.0f;
    h.prim = 0;
    hit_set_prims(sc, sc.world, r, h, 0u, st, tc);
}
__device__ __forceinline__ void hit_world_insts(const DevScene &sc, TravRay &r, HitState &h, LaneStack &st, TravCounters &tc) {
    traverse(sc, sc.world.insts, r, st, tc, [&](uint32_t gi) { hit_instance(sc, gi, r, h, st, tc); });
}

}  // namespace ptd
