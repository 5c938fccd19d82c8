// Persistent-wave closest-hit search over the world's triangle, sphere and quad buffers
// (the first three lines of PrimitiveBuffer::hit, src/objects.rs:192-194) on 64-byte
// two-box nodes.
//
// Every ray runs exactly the sequence of box and primitive tests of traverse<true> in
// dev_traverse.h (hence of InnerPrimitiveBuffer::traverse, src/bvh.rs:725-797); what
// changes is which lane runs which ray and when.  Rays are not bound to lanes in advance:
// a wave claims pieces of the ray queue through an atomic cursor (a whole 256-entry chunk
// per atomic when the queue is long, 64 entries when it is short, so that short queues
// still spread over many waves), and deals the claimed entries to its idle lanes.  When
// fewer than `refill_below` of its lanes still have a ray in flight it leaves the traversal
// loop, hands the finished rays to the sink and refills the idle lanes; the other lanes
// keep their whole traversal state (node reference, stack, buffer in progress) across that
// round trip.  A wave ends when the queue is used up and its lanes are done.
//
// Inside the traversal loop every pass gives each lane one step of its own sequence: one
// interior node (A), or one primitive of the leaf it has reached (B).  Lanes at a leaf wait
// until `leaf_batch` of them are (or no lane has a node left), so that the primitive test
// runs with a fair share of the wave; nothing else about a lane's sequence depends on what
// the other lanes do.
#pragma once
#include "dev_traverse.h"

namespace ptd {

constexpr uint32_t REF_NONE = 0xFFFFFFFFu;    // no node pending: pop, or go to the next buffer
constexpr uint32_t REF_LINEAR = 0xFFFFFFFEu;  // buffer without a BVH: all primitives in order (src/bvh.rs:726-737)
// Neither value can be a leaf reference: both decode to 127 primitives at an offset of at
// least 2^24 - 2, past the largest offset a leaf may have.

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

// source(i, slot, ray): loads queue entry i, false when it is padding.   sink(done, slot,
// ray, hit): called by all 64 lanes of the wave together (so it may use ballots), done ==
// true on lanes whose ray has just been finished.  `n_chunks` chunks of the queue are in
// use; `cursor` counts the 64-entry pieces handed out.
template <class Source, class Sink>
__device__ __forceinline__ void walk_world_prims(const DevScene &sc, uint32_t n_chunks, uint32_t *cursor, int refill_below, int leaf_batch, LaneStack &st,
                                                 Source &&source, Sink &&sink) {
    const DevSet &set = sc.world;
    constexpr uint32_t PIECES = QUEUE_CHUNK / 64u;
    const uint32_t n_pieces = n_chunks * PIECES;
    const uint32_t claim = n_chunks >= 2u * wave_count() ? PIECES : 1u;  // pieces per atomic
    // the cursors live in memory every XCD shares: waves that cannot get a piece do not poll them
    if ((uint64_t)wave_index() * claim >= n_pieces) return;
    uint32_t res_next = 0u, res_end = 0u;  // claimed entries not dealt yet (same on all lanes)
    bool used_up = false;
    // phase: 0..2 = buffer being searched, 3 = finished (sink pending), 4 = lane has no ray
    // cur: the node to visit next, or the leaf whose primitives [li, lend) are pending, or REF_NONE
    uint32_t phase = 4u, cur = REF_NONE, slot = 0u, node_base = 0u, prim_base = 0u, li = 0u, lend = 0u;
    TravRay r = make_trav_ray(mk3(0.0f, 0.0f, 0.0f), mk3(1.0f, 1.0f, 1.0f), 0.0f, 0.0f);
    HitState h;
    h.code = PTD_HIT_MISS;
    h.u = h.v = 0.0f;
    h.prim = 0u;
    TravCounters tc{0, 0, 0, 0};
    auto open_leaf = [&]() {
        if ((cur & WIDE_LEAF) && cur != REF_NONE) {
            if (cur == REF_LINEAR) {
                li = prim_base;
                lend = li + (phase == 0u ? set.tris.prim_count : (phase == 1u ? set.spheres.prim_count : set.quads.prim_count));
            } else {
                li = prim_base + (cur & 0xFFFFFFu);
                lend = li + ((cur >> 24) & 0x7Fu);
            }
        }
    };
    for (;;) {
        // ---- deal new rays to idle lanes
        for (;;) {
            const unsigned long long idle = __ballot(phase == 4u);
            if (idle == 0ull) break;
            if (res_next == res_end) {
                if (used_up) break;
                uint32_t got = 0u;
                if (lane_id() == 0u) got = atomicAdd(cursor, claim);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (got >= n_pieces) {
                    used_up = true;
                    break;
                }
                res_next = got * 64u;
                res_end = (got + claim < n_pieces ? got + claim : n_pieces) * 64u;
            }
            const uint32_t want = (uint32_t)__popcll(idle), avail = res_end - res_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (phase == 4u && rank < avail && source(res_next + rank, slot, r)) {
                h.code = PTD_HIT_MISS;
                h.u = h.v = 0.0f;
                h.prim = 0u;
                phase = 0u;
                node_base = set.tris.wide_first;
                prim_base = set.tris.prim_first;
                cur = walk_begin(set.tris, r);
                open_leaf();
            }
            res_next += want < avail ? want : avail;
        }
        if (__ballot(phase < 3u) == 0ull) break;  // nothing in flight, so nothing left to claim either

        // ---- traverse until too few lanes are busy
        const bool can_refill = !used_up || res_next != res_end;
        // min/max form of the box test when no ray of the wave can produce a NaN in it
        const bool plain = sc.boxes_finite != 0u && __ballot(phase < 3u && !ray_is_plain(r)) == 0ull;
        while (phase < 3u) {
            // (A) one interior node, for the lanes that have one pending
            if (!(cur & WIDE_LEAF)) {
                const float4 *w = sc.wide + 4 * (size_t)(node_base + cur);
                const float4 a = w[0], b = w[1], c = w[2], d = w[3];
                float el, er;
                bool hl, hr;
                if (plain) {
                    hl = slab_test_entry_plain(a.x, a.y, a.z, a.w, b.x, b.y, r, el);
                    hr = slab_test_entry_plain(b.z, b.w, c.x, c.y, c.z, c.w, r, er);
                } else {
                    hl = slab_test_entry(a.x, a.y, a.z, a.w, b.x, b.y, r, el);
                    hr = slab_test_entry(b.z, b.w, c.x, c.y, c.z, c.w, r, er);
                }
                const uint32_t lref = __float_as_uint(d.x), rref = __float_as_uint(d.y), axis = __float_as_uint(d.z);
                const float da = (axis == 0) ? r.d.x : ((axis == 1) ? r.d.y : r.d.z);
                const bool neg = da < 0.0f;
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
                open_leaf();
            }
            // (B) one primitive of a pending leaf, once enough lanes have one or no lane has a node left
            {
                const unsigned long long at_leaf = __ballot(li < lend), at_node = __ballot(!(cur & WIDE_LEAF));
                if (at_leaf != 0ull && (at_node == 0ull || __popcll(at_leaf) >= leaf_batch)) {
                    if (li < lend) {
                        if (phase == 0u) hit_prim<PT_PRIM_TRI>(sc, li, r, h, 0u, tc);
                        else if (phase == 1u) hit_prim<PT_PRIM_SPHERE>(sc, li, r, h, 0u, tc);
                        else hit_prim<PT_PRIM_QUAD>(sc, li, r, h, 0u, tc);
                        li++;
                        if (li == lend) {
                            cur = walk_pop(st, r);
                            open_leaf();
                        }
                    }
                }
            }
            // (C) nothing left in this buffer (its stack is empty): on to the next one
            if (cur == REF_NONE) {
                phase++;
                if (phase == 1u) {
                    node_base = set.spheres.wide_first;
                    prim_base = set.spheres.prim_first;
                    cur = walk_begin(set.spheres, r);
                    open_leaf();
                } else if (phase == 2u) {
                    node_base = set.quads.wide_first;
                    prim_base = set.quads.prim_first;
                    cur = walk_begin(set.quads, r);
                    open_leaf();
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
