// Persistent-wave closest-hit search over the world's triangle, sphere and quad buffers
// (the first three lines of PrimitiveBuffer::hit, src/objects.rs:192-194) on 64-byte
// two-box nodes.
//
// Every ray runs exactly the sequence of box and primitive tests of traverse<true> in
// dev_traverse.h (hence of InnerPrimitiveBuffer::traverse, src/bvh.rs:725-797); what
// changes is which lane runs which ray and when:
//   - rays are not bound to lanes in advance.  A wave draws rays from the queue through an
//     atomic cursor, and whenever fewer than `refill_below` of its lanes still have a ray in
//     flight it leaves the traversal loop, hands the finished rays to the sink and gives
//     the idle lanes new rays.  Lanes keep their whole traversal state (node reference,
//     stack, buffer in progress) in registers/LDS across that round trip;
//   - the queue is cut into eight contiguous regions with one cursor each; a wave starts on
//     the region of blockIdx.x % 8 (workgroups with equal residue are observed to share an
//     XCD, hence an L2) and moves on to the other regions when its own is used up.  Only
//     speed depends on that placement: every entry is drawn exactly once whatever it is,
//     and a wave ends when all eight regions are used up and its lanes are done.
#pragma once
#include "dev_traverse.h"

namespace ptd {

constexpr uint32_t WALK_REGIONS = 8;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;    // no node pending: pop, or go to the next buffer
constexpr uint32_t REF_LINEAR = 0xFFFFFFFEu;  // buffer without a BVH: all primitives in order (src/bvh.rs:726-737)
// Neither value can be a leaf reference: both decode to 127 primitives at an offset of at
// least 2^24 - 2, past the largest offset a leaf may have.

struct WalkCursor {
    uint32_t next[WALK_REGIONS];  // entries of each region handed out so far
};


// First node of a buffer for ray r: the buffer's own early-outs and the root's box.
__device__ __forceinline__ uint32_t walk_begin(const DevBuffer &buf, const TravRay &r) {
    if (buf.prim_count == 0) return REF_NONE;
    if (buf.node_count == 0) return REF_LINEAR;
    float e;
    if (!slab_test_entry(buf.rmin[0], buf.rmin[1], buf.rmin[2], buf.rmax[0], buf.rmax[1], buf.rmax[2], r, e)) return REF_NONE;
    return buf.root_ref;
}

// Next stacked node whose entry distance is still within reach, REF_NONE when the stack is empty.
__device__ __forceinline__ uint32_t walk_pop(LaneStack &st, const TravRay &r) {
    uint32_t ref;
    float e;
    for (;;) {
        if (!st.pop(0, ref, e)) return REF_NONE;
        if (!(r.t_max < e)) return ref;
    }
}

// source(i, slot, ray): loads queue entry i.   sink(done, slot, ray, hit): called by all 64
// lanes of the wave together (so it may use ballots), done == true on lanes whose ray has
// just been finished.
template <class Source, class Sink>
__device__ __forceinline__ void walk_world_prims(const DevScene &sc, uint32_t n, uint32_t *cursor, int refill_below, LaneStack &st,
                                                 Source &&source, Sink &&sink) {
    const DevSet &set = sc.world;
    // A short queue is one region, and only as many waves as it has 64-entry pieces take part:
    // the cursors live in memory every XCD shares, and the fewer waves poll them the better.
    const uint32_t n_regions = n >= WALK_REGIONS * 8192u ? WALK_REGIONS : 1u;
    if ((uint64_t)(blockIdx.x * (BLOCK / 64u) + threadIdx.x / 64u) * 64u >= n) return;
    const uint32_t seg = ((n + n_regions * 64u - 1u) / (n_regions * 64u)) * 64u;  // region length
    uint32_t region = blockIdx.x % n_regions, regions_left = n_regions;
    // phase: 0..2 = buffer being searched, 3 = finished (sink pending), 4 = lane has no ray
    uint32_t phase = 4u, cur = REF_NONE, slot = 0u, node_base = 0u, prim_base = 0u;
    TravRay r = make_trav_ray(mk3(0.0f, 0.0f, 0.0f), mk3(1.0f, 1.0f, 1.0f), 0.0f, 0.0f);
    HitState h;
    h.code = PTD_HIT_MISS;
    h.u = h.v = 0.0f;
    h.prim = 0u;
    TravCounters tc{0, 0, 0, 0};
    for (;;) {
        // ---- hand new rays to idle lanes
        while (regions_left != 0u) {
            const unsigned long long idle = __ballot(phase == 4u);
            if (idle == 0ull) break;
            const uint32_t want = (uint32_t)__popcll(idle);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const int leader = __ffsll((long long)idle) - 1;
            const uint64_t first = (uint64_t)region * seg;
            const uint32_t len = first >= n ? 0u : (uint32_t)((n - first) < seg ? (n - first) : seg);
            uint32_t base = len;
            if (len != 0u) {
                uint32_t got = 0u;
                if ((int)lane_id() == leader) got = atomicAdd(&cursor[region], want);
                base = (uint32_t)__builtin_amdgcn_readlane((int)got, leader);
            }
            if (phase == 4u && base < len && rank < len - base) {
                source((uint32_t)first + base + rank, slot, r);
                h.code = PTD_HIT_MISS;
                h.u = h.v = 0.0f;
                h.prim = 0u;
                phase = 0u;
                node_base = set.tris.wide_first;
                prim_base = set.tris.prim_first;
                cur = walk_begin(set.tris, r);
            }
            if (base >= len || want >= len - base) {  // region used up
                region = (region + 1u) % n_regions;
                regions_left--;
            }
        }
        if (__ballot(phase < 3u) == 0ull) break;  // nothing in flight: the queue is used up too (or every draw came up empty)

        // ---- traverse until too few lanes are busy
        const bool can_refill = regions_left != 0u;
        while (phase < 3u) {
            const bool neg_x = r.d.x < 0.0f, neg_y = r.d.y < 0.0f, neg_z = r.d.z < 0.0f;
            // interior nodes: descend until a leaf (or nothing) is pending
            while (!(cur & WIDE_LEAF)) {
                const float4 *w = sc.wide + 4 * (size_t)(node_base + cur);
                const float4 a = w[0], b = w[1], c = w[2], d = w[3];
                float el, er;
                const bool hl = slab_test_entry(a.x, a.y, a.z, a.w, b.x, b.y, r, el);
                const bool hr = slab_test_entry(b.z, b.w, c.x, c.y, c.z, c.w, r, er);
                const uint32_t lref = __float_as_uint(d.x), rref = __float_as_uint(d.y), axis = __float_as_uint(d.z);
                const bool neg = (axis == 0) ? neg_x : ((axis == 1) ? neg_y : neg_z);
                const uint32_t near_ref = neg ? rref : lref, far_ref = neg ? lref : rref;
                const bool hn = neg ? hr : hl, hf = neg ? hl : hr;
                const float ef = neg ? el : er;
                if (hn) {
                    if (hf) st.push(far_ref, ef);
                    cur = near_ref;
                } else if (hf) {
                    cur = far_ref;
                } else {
                    cur = walk_pop(st, r);
                }
            }
            if (cur != REF_NONE) {
                // a leaf: its primitives in order
                uint32_t i = prim_base, end;
                if (cur == REF_LINEAR) {
                    end = i + (phase == 0u ? set.tris.prim_count : (phase == 1u ? set.spheres.prim_count : set.quads.prim_count));
                } else {
                    i += cur & 0xFFFFFFu;
                    end = i + ((cur >> 24) & 0x7Fu);
                }
                if (phase == 0u) {
#pragma unroll 1
                    for (; i < end; i++) hit_prim<PT_PRIM_TRI>(sc, i, r, h, 0u, tc);
                } else if (phase == 1u) {
#pragma unroll 1
                    for (; i < end; i++) hit_prim<PT_PRIM_SPHERE>(sc, i, r, h, 0u, tc);
                } else {
#pragma unroll 1
                    for (; i < end; i++) hit_prim<PT_PRIM_QUAD>(sc, i, r, h, 0u, tc);
                }
                cur = walk_pop(st, r);
            } else {
                // this buffer is done (its stack is empty): on to the next one
                phase++;
                if (phase == 1u) {
                    node_base = set.spheres.wide_first;
                    prim_base = set.spheres.prim_first;
                    cur = walk_begin(set.spheres, r);
                } else if (phase == 2u) {
                    node_base = set.quads.wide_first;
                    prim_base = set.quads.prim_first;
                    cur = walk_begin(set.quads, r);
                }
            }
            if (can_refill && __popcll(__ballot(phase < 3u)) < refill_below) break;
        }

        // ---- finished rays leave
        sink(phase == 3u, slot, r, h);
        if (phase == 3u) phase = 4u;
    }
}

}  // namespace ptd
