// Closest-hit search for a wave whose 64 rays are nearly the same ray (camera rays of neighbouring
// pixels and strata): the reference's loop (InnerPrimitiveBuffer::traverse, src/bvh.rs:725-797) run once
// per WAVE over the 32-byte reference-order nodes, with a lane mask saying whose loop is at this node.
//
// Every lane's own sequence of box and primitive tests is exactly the sequence traverse() in
// dev_traverse.h performs for its ray, each test with that lane's t_max of the moment:
//   - a node is visited with the mask M of the lanes whose loop would visit it now; each of them tests
//     the node's box itself (the reference tests a box when it visits the node, never earlier);
//   - at an interior node the lanes that passed go on to the child their own direction makes the near
//     one.  When they all agree the far child is stacked with the same mask.  When they do not (A: left
//     first, B: right first) the wave visits left with A, then right with A and B, then left with B:
//     each lane still sees its two children in its own order, and what the wave does in between does
//     not touch it;
//   - at a leaf the lanes that passed test its primitives in order.
// Nodes and primitives are the same for all lanes, so they are read through the constant address space
// (scalar loads, one per wave), and the stack holds one (node, mask) pair per entry for the whole wave.
#pragma once
#include "dev_traverse.h"

namespace ptd {

// stack entries per wave: two per level in the worst case (lanes that disagree about the near child), trees
// are at most 64 levels deep (checked at upload), and the tree of an instanced set is searched with the
// entries of the world's instance tree still in place below it
constexpr int PACKET_LEVELS = 2 * (2 * 64 + 4);
__shared__ uint32_t g_packet_stack[BLOCK / 64][PACKET_LEVELS][3];

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// A pointer to memory that nothing writes while the scene lives, whose loads at a wave-uniform address go over
// the scalar unit (constant address space): one load per wave, the values arrive in scalar registers.
// (The alternative, one coalesced vector load per wave with lane k fetching dword k and v_readlane to spread
// the values, was measured slower: 0.099 ns per camera ray against 0.087, one ray per lane taking 0.106.)
typedef float v4f __attribute__((ext_vector_type(4)));
struct ScalarF4 {
    const __attribute__((address_space(4))) v4f *p;
    __device__ __forceinline__ float4 operator[](uint32_t i) const {
        const v4f v = p[i];
        return make_float4(v.x, v.y, v.z, v.w);
    }
};
__device__ __forceinline__ ScalarF4 scalar_f4(const float4 *q) { return ScalarF4{(const __attribute__((address_space(4))) v4f *)q}; }

// leaf(g, on, sp): primitive g (the same in every lane) for the lanes with on == true; sp = the wave's stack
// entries in use at that moment (a leaf that searches another tree starts its own above them).
// `plain`: no lane of the wave can produce a NaN in the box test (ray_is_plain), finite boxes.
// `floor`: stack entries below it belong to an enclosing search.
template <class LeafFn>
__device__ __forceinline__ void packet_traverse(const DevScene &sc, const DevBuffer &buf, unsigned long long active, bool plain, TravRay &r, int floor,
                                                LeafFn &&leaf) {
    if (buf.prim_count == 0u || active == 0ull) return;
    // Lane masks stay in scalar registers: which lanes a mask names is read back as a condition
    // (inverse ballot: the mask itself becomes the select / branch condition, no vector instruction), and masks are
    // combined with scalar instructions; the kernel is bound by the number of vector instructions it issues.
    if (buf.node_count == 0u) {  // no BVH: src/bvh.rs:726-737
        const bool on = __builtin_amdgcn_inverse_ballot_w64(active);
#pragma unroll 1
        for (uint32_t i = 0; i < buf.prim_count; i++) leaf(buf.prim_first + i, on, floor);
        return;
    }
    const ScalarF4 nodes = scalar_f4(sc.nodes + 2 * (size_t)buf.node_first);
    uint32_t(*stack)[3] = g_packet_stack[threadIdx.x / 64u];
    int sp = floor;
    uint32_t cur = 0u;
    unsigned long long mask = active;
    // which lanes go against each axis: three masks for the whole search.  They choose the near child (src/bvh.rs:757)
    // with scalar instructions (per visit this was two compares and two selects to pick the direction's component by
    // the node's axis, a third compare and a ballot), and in the plain form of the box test the near and far plane of
    // each axis
    const unsigned long long neg_x = __ballot(r.d.x < 0.0f), neg_y = __ballot(r.d.y < 0.0f), neg_z = __ballot(r.d.z < 0.0f);
    for (;;) {
        const float4 n0 = nodes[2u * cur], n1 = nodes[2u * cur + 1u];
        // every lane runs the test, the ones outside `mask` on whatever their registers hold (it reads no memory)
        float e;
        const unsigned long long passed =
            (plain ? __ballot(slab_test_entry_signed(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, r, __builtin_amdgcn_inverse_ballot_w64(neg_x),
                                                     __builtin_amdgcn_inverse_ballot_w64(neg_y), __builtin_amdgcn_inverse_ballot_w64(neg_z), e))
                   : __ballot(slab_test(n0, n1, r))) &
            mask;
        bool descend = false;
        if (passed != 0ull) {
            const uint32_t a = __float_as_uint(n1.z), b = __float_as_uint(n1.w);
            if (!(b & PT_NODE_LEAF)) {
                const unsigned long long right_first = (b == 0u ? neg_x : (b == 1u ? neg_y : neg_z)) & passed, left_first = passed & ~right_first;
                auto push = [&](uint32_t node, unsigned long long m) {
                    if (lane_id() == 0u) {
                        stack[sp][0] = node;
                        stack[sp][1] = (uint32_t)m;
                        stack[sp][2] = (uint32_t)(m >> 32);
                    }
                    sp++;
                };
                if (right_first == 0ull) {
                    push(a, left_first);
                    cur = cur + 1u;
                    mask = left_first;
                } else if (left_first == 0ull) {
                    push(cur + 1u, right_first);
                    cur = a;
                    mask = right_first;
                } else {
                    push(cur + 1u, right_first);
                    push(a, passed);
                    cur = cur + 1u;
                    mask = left_first;
                }
                descend = true;
            } else {
                const uint32_t cnt = b & 0xFFFFu;
                const bool hit = __builtin_amdgcn_inverse_ballot_w64(passed);
#pragma unroll 1
                for (uint32_t i = 0; i < cnt; i++) leaf(buf.prim_first + a + i, hit, sp);
            }
        }
        if (!descend) {
            if (sp == floor) break;
            sp--;
            cur = uniform(stack[sp][0]);
            mask = (unsigned long long)uniform(stack[sp][1]) | ((unsigned long long)uniform(stack[sp][2]) << 32);
        }
    }
}

// The triangle, sphere and quad buffers of one PrimitiveBuffer, in that order (src/objects.rs:192-194), for a
// wave of rays; lanes outside `active` have no ray.
__device__ __forceinline__ void packet_hit_set(const DevScene &sc, const DevSet &set, unsigned long long active, bool plain, TravRay &r, HitState &h,
                                               uint32_t inst_code, int floor) {
    const ScalarF4 tri_hot = scalar_f4(sc.tri_hot), spheres = scalar_f4(sc.spheres), quads = scalar_f4(sc.quads);
    packet_traverse(sc, set.tris, active, plain, r, floor, [&](uint32_t g, bool on, int) {
        const float4 a = tri_hot[3u * g], b = tri_hot[3u * g + 1u], c = tri_hot[3u * g + 2u];
                if (on) hit_tri_flat(a, b, c, g, r, h, inst_code);
    });
    packet_traverse(sc, set.spheres, active, plain, r, floor, [&](uint32_t g, bool on, int) {
        const float4 s = spheres[g];
        if (on) hit_sphere(s, g, r, h, inst_code);
    });
    packet_traverse(sc, set.quads, active, plain, r, floor, [&](uint32_t g, bool on, int) {
        const float4 q0 = quads[4u * g], q1 = quads[4u * g + 1u], q2 = quads[4u * g + 2u], q3 = quads[4u * g + 3u];
        if (on) hit_quad(q0, q1, q2, q3, g, r, h, inst_code);
    });
}

// PrimitiveBuffer::hit on the world (src/objects.rs:191-197): its own three buffers, then its instance buffer, where
// a "primitive" is an instance: the lanes that reach it take their rays into its frame (Instance::hit,
// src/objects.rs:384-401: direction not renormalised, so t is shared between the frames), the wave searches the
// instanced set's buffers for them above the stack entries of the instance tree, and a hit in there replaces
// the lane's hit and shortens its ray as the reference's local HitRecord does when it is copied out.
// INSTANCES: the world has an instance buffer (the code for it costs a third of the registers).
template <bool INSTANCES>
__device__ __forceinline__ void packet_hit_world(const DevScene &sc, unsigned long long active, bool plain, TravRay &r, HitState &h) {
    packet_hit_set(sc, sc.world, active, plain, r, h, 0u, 0);
    if constexpr (!INSTANCES) return;
    const ScalarF4 insts = scalar_f4(sc.insts);
    packet_traverse(sc, sc.world.insts, active, plain, r, 0, [&](uint32_t gi, bool on, int sp) {
        Affine inv;
        const float4 r0 = insts[7u * gi], r1 = insts[7u * gi + 1u], r2 = insts[7u * gi + 2u], last = insts[7u * gi + 6u];
        inv.m[0] = r0.x, inv.m[1] = r0.y, inv.m[2] = r0.z, inv.m[3] = r0.w;
        inv.m[4] = r1.x, inv.m[5] = r1.y, inv.m[6] = r1.z, inv.m[7] = r1.w;
        inv.m[8] = r2.x, inv.m[9] = r2.y, inv.m[10] = r2.z, inv.m[11] = r2.w;
        const uint32_t child = uniform(__float_as_uint(last.x));
        TravRay local = make_trav_ray(xf_point(inv, r.o), xf_vector(inv, r.d), r.t_min, r.t_max);
        HitState hl;
        hl.code = PTD_HIT_MISS;
        hl.u = hl.v = 0.0f;
        hl.prim = 0u;
        const unsigned long long in = __ballot(on);
        const bool plain_local = sc.boxes_finite != 0u && __ballot(on && !ray_is_plain(local)) == 0ull;
        packet_hit_set(sc, sc.sets[child], in, plain_local, local, hl, (gi - sc.world.insts.prim_first + 1u) << 2, sp);
        if (on && hl.code != PTD_HIT_MISS) {
            h = hl;
            r.t_max = local.t_max;
        }
    });
}

}  // namespace ptd
