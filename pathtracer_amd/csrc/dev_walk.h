// Persistent-wave closest-hit search: PrimitiveBuffer::hit (src/objects.rs:191-197) over the
// world's triangle, sphere and quad buffers and, when INSTANCES, its instance buffer with
// Instance::hit (src/objects.rs:384-401) into the instanced sets, on 64-byte two-box nodes.
//
// Nodes.  A 64-byte node holds both children's boxes.  Both are tested when the parent is
// visited (nothing is intersected between that moment and the reference's own test of the
// near child, so t_max is the same); a far child that passes is pushed with its entry
// distance E and is dropped at pop time when t_max has shrunk below E, which is the outcome
// of the reference's lazy test of that box at that moment (slab_test_entry in
// dev_traverse.h).  The root's own box comes with the buffer descriptor.  Every ray
// therefore visits the leaves InnerPrimitiveBuffer::traverse (src/bvh.rs:725-797) visits,
// in its order, with its t_max, and tests the same primitives: ids and t are bit-identical
// to traverse() in dev_traverse.h, ties included.
//
// Scheduling.  What changes is which lane runs which ray and when.  Rays are not bound to lanes in advance:
// a wave claims pieces of the ray queue through an atomic cursor (a whole 256-entry chunk
// per atomic when the queue is long, 64 entries when it is short, so that short queues
// still spread over many waves), and deals the claimed entries to its idle lanes.  When
// fewer than `refill_below` of its lanes still have a ray in flight it leaves the traversal
// loop, hands the finished rays to the sink and refills the idle lanes; the other lanes
// keep their whole traversal state (node reference, stack, buffer in progress) across that
// round trip.  A wave ends when the queue is used up and its lanes are done.
//
// Inside the traversal loop every pass gives each lane a step of its own sequence: up to two
// interior nodes in a row (A), or one primitive of the leaf it has reached (B).  Lanes at a leaf wait
// until `leaf_batch` of them are (or no lane has a node left), so that the primitive test
// runs with a fair share of the wave; nothing else about a lane's sequence depends on what
// the other lanes do.
//
// Instances.  A "primitive" of the world's instance buffer is an instance: the lane takes
// its ray into the instance's frame (direction not renormalised, so t is shared between
// the frames), leaves a return record on its stack and searches the three buffers of the
// instanced set like the world's, with the instance tag in the hit code.  A hit inside the
// instance replaces the ray's hit and shrinks its t_max exactly as the reference's local
// HitRecord does when it is copied out (objects.rs:395-399), so the hit and t_max are
// updated in place.  When the set's last buffer is exhausted the lane reloads its
// world-frame ray and goes on with the next instance of the leaf it came from.
#pragma once
#include "dev_traverse.h"

namespace ptd {

// A build with PT_WALK_STATS counts, per launch-wide wave pass, how many lanes took part in each kind of step
// (tools/walk_stats.py reads the totals); nothing of it is in the normal build.
#ifdef PT_WALK_STATS
__device__ unsigned long long g_walk_stats[16];
#endif

// PT_TOP_LDS = N > 0: the first N nodes of the world's triangle tree (its top, breadth-first) are copied
// to LDS by every workgroup and read from there.  Off by default: measured 6 % slower with N = 64 and 17 %
// slower with N = 255 on a 184 k-triangle scene; the L1 already serves those nodes and the kernel is bound
// by instruction issue, where the extra compare and branch per node cost more than the shorter latency saves.
#ifndef PT_TOP_LDS
#define PT_TOP_LDS 0
#endif
#if PT_TOP_LDS > 0
__shared__ float4 g_top_nodes[4 * PT_TOP_LDS];
#endif

// Stack / `cur` values that are not node references.  None can be a leaf reference: all
// decode to 127 primitives at an offset of at least 2^24 - 4, past the largest offset a
// leaf may have.
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;    // nothing pending: the stack is empty
constexpr uint32_t REF_LINEAR = 0xFFFFFFFEu;  // buffer without a BVH: all primitives in order (src/bvh.rs:726-737)
constexpr uint32_t REF_RETURN = 0xFFFFFFFDu;  // bottom of an instance's part of the stack; the record below it
                                              // holds (leaf of the instance buffer in progress, next instance)
constexpr uint32_t REF_POP = 0xFFFFFFFCu;     // `cur` only: the lane needs the next entry of its stack
constexpr uint32_t PHASE_INSTS = 3u, PHASE_CHILD = 4u, PHASE_DONE = 7u, PHASE_IDLE = 8u;

// A buffer that needs no search: empty, without a tree, or with a tree that is one leaf.
__host__ __device__ inline bool walk_one_leaf(const DevBuffer &b) {
    return b.prim_count == 0u || b.tree == 0u || (b.tree == 1u && (b.root_ref & WIDE_LEAF) != 0u && b.root_ref < REF_POP);
}
// walk_world's TAIL form applies to a world without instances whose sphere and quad buffers are of that kind.
__host__ __device__ inline bool walk_tail_applies(const DevSet &world) { return world.insts.prim_count == 0u && walk_one_leaf(world.spheres) && walk_one_leaf(world.quads); }

// First node of a buffer for ray r: the buffer's own early-outs and the root's box.
__device__ __forceinline__ uint32_t walk_begin(const DevBuffer &buf, const TravRay &r) {
    if (buf.prim_count == 0) return REF_NONE;
    if (buf.tree == 0u) return REF_LINEAR;
    float e;
    if (!slab_test_entry(buf.rmin[0], buf.rmin[1], buf.rmin[2], buf.rmax[0], buf.rmax[1], buf.rmax[2], r, e)) return REF_NONE;
    return buf.root_ref;
}

// Next stacked node whose entry distance is still within reach; REF_RETURN when the
// entries of the instance in progress are used up (that record stays on the stack until the
// lane leaves the instance), REF_NONE when the stack is empty, REF_POP when the two entries looked at were both out
// of reach (the lane goes on popping in the wave's next pass: a loop here would be a loop of the whole wave for the
// sake of the few lanes that pop, with the scalar bookkeeping of a divergent loop).
template <bool INSTANCES>
__device__ __forceinline__ uint32_t walk_pop(LaneStack &st, const TravRay &r) {
    uint32_t out = REF_POP;
#pragma unroll
    for (int attempt = 0; attempt < 2; attempt++) {
        if (out == REF_POP) {
            uint32_t ref;
            float e;
            if (!st.pop_flat(ref, e)) {
                out = REF_NONE;
            } else if (INSTANCES && ref == REF_RETURN) {
                st.push_flat(ref, e);
                out = ref;
            } else if (!(r.t_max < e)) {
                out = ref;
            }
        }
    }
    return out;
}

// A build with PT_BOUNDS_CHECK verifies every node, primitive, instance and set index before
// it is used; a bad one is counted and ends that ray instead of reading out of range.
#ifdef PT_BOUNDS_CHECK
// ... and a lane that is still not done after WALK_MAX_PASSES passes leaves its state behind
// (phase, cur, li, lend, instance tag, stack depth) and gives up.
constexpr uint32_t WALK_MAX_PASSES = 1u << 20;
#define WALK_CHECK(cond, id)                                                              \
    if (!(cond)) {                                                                        \
        if (atomicAdd(bad_refs, 1ull) == 0ull) {                                          \
            bad_refs[2] = phase | ((unsigned long long)(id) << 32);                       \
            bad_refs[3] = cur | ((unsigned long long)node_base << 32);                    \
            bad_refs[4] = ((unsigned long long)li << 32) | lend;                          \
            bad_refs[5] = inst_code | ((unsigned long long)prim_base << 32);              \
            bad_refs[6] = (unsigned long long)st.sp | (st.has_top() ? 1ull << 32 : 0ull);   \
            bad_refs[7] = slot | ((unsigned long long)child << 32);                       \
        }                                                                                 \
        phase = PHASE_DONE;                                                               \
        cur = REF_NONE;                                                                   \
        li = lend = 0u;                                                                   \
        st.clear();                                                                       \
        continue;                                                                         \
    }
#else
#define WALK_CHECK(cond, id)
#endif

// source(i, slot, ray): loads queue entry i, false when it is padding.
// reload(slot, o, d): the world-frame origin and direction of that ray again (INSTANCES only).
// sink(done, slot, ray, hit): called by all 64 lanes of the wave together (so it may use
// ballots), done == true on lanes whose ray has just been finished; ray.t_max is the hit
// distance.  The work is `n_pieces` pieces of 64 entries; `cursor`: WORK_LISTS control words,
// CTL_STRIDE apart, zero before the launch, that count the blocks of pieces handed out.  `bad_refs` is only touched in the bounds-checking build.
// ANY_HIT: a ray is finished by the first primitive it hits (shadow rays: is there anything in [t_min, t_max]?).
// TAIL: the caller has checked that the world has no instances and that its sphere and quad buffers are of the kind
// described below (walk_tail_applies); the code for them is then compiled in, and left out otherwise (it costs the
// other scenes registers: 4 to 5 % of their rate when it was always there).
template <bool INSTANCES, bool ANY_HIT = false, bool TAIL = false, class Source, class Reload, class Sink>
__device__ __forceinline__ void walk_world(const DevScene &sc, uint32_t n_pieces, uint32_t *cursor, int refill_below, int leaf_batch,
                                           LaneStack &st, unsigned long long *bad_refs, Source &&source, Reload &&reload, Sink &&sink) {
    const DevSet &set = sc.world;
#if PT_TOP_LDS > 0
    const uint32_t top_base = set.tris.wide_first;
    const uint32_t top_n = set.tris.tree != 0u && sc.n_wide - top_base >= (uint32_t)PT_TOP_LDS ? (uint32_t)PT_TOP_LDS : 0u;
    for (uint32_t i = threadIdx.x; i < 4u * top_n; i += blockDim.x) g_top_nodes[i] = sc.wide[4 * (size_t)top_base + i];
    __syncthreads();
#endif
    // Sphere and quad buffers whose tree is one leaf (or that are empty, or have no tree) need no search: once a lane is
    // through the triangles its ray is finished as far as the loop below is concerned, and what PrimitiveBuffer::hit
    // does next, the root's box and the leaf's primitives in order, spheres before quads, is done for all the rays that
    // leave together, just before they are handed on.  In the loop those few tests took a lane four or five passes, in
    // primitive steps of their own kind with a handful of lanes.
    constexpr bool tail_after = TAIL && !INSTANCES;
    constexpr uint32_t PIECES = QUEUE_CHUNK / 64u;
    const uint32_t claim = n_pieces >= 2u * PIECES * wave_count() ? PIECES : 1u;  // pieces per atomic
    // the cursors live in memory every XCD shares: waves that cannot get a piece do not poll them
    if ((uint64_t)wave_index() * claim >= n_pieces) return;
    // blocks of `claim` pieces, block b on list b mod WORK_LISTS, a wave drawing from the list of its XCD first: one
    // counter for the whole chip completes an atomic with return every 11 ns whatever the number of waves, which a
    // launch of a few hundred thousand rays spends a third of its time waiting for
    BlockClaim blocks;
    blocks.init(cursor, (n_pieces + claim - 1u) / claim);
    uint32_t res_next = 0u, res_end = 0u;  // claimed entries not dealt yet (same on all lanes)
    bool used_up = false;
    // phase: 0..2 = world buffer being searched, 3 = the world's instance buffer, 4..6 = buffer of
    // the instanced set, PHASE_DONE = finished (sink pending), PHASE_IDLE = lane has no ray
    // cur: the node to visit next, or the leaf whose primitives [li, lend) are pending, or REF_*
    uint32_t phase = PHASE_IDLE, cur = REF_NONE, slot = 0u, node_base = 0u, prim_base = 0u, li = 0u, lend = 0u;
    uint32_t inst_code = 0u, child = 0u;  // (instance + 1) << 2 and its set while inside an instance
    bool lane_plain = true;               // the min/max box test is exact for this lane's current ray
    TravRay r = make_trav_ray(mk3(0.0f, 0.0f, 0.0f), mk3(1.0f, 1.0f, 1.0f), 0.0f, 0.0f);
    HitState h;
    h.code = PTD_HIT_MISS;
    h.u = h.v = 0.0f;
    h.prim = 0u;
    TravCounters tc{0, 0, 0, 0};
    (void)bad_refs;
#ifdef PT_BOUNDS_CHECK
    uint32_t passes = 0u, rounds = 0u;
#endif
    // [li, lend) of the leaf in `cur`; `count` = primitives of the buffer in progress (for REF_LINEAR)
    auto open_leaf = [&](uint32_t count) {
        if ((cur & WIDE_LEAF) && cur < REF_POP) {
            li = prim_base + (cur & 0xFFFFFFu);
            lend = li + ((cur >> 24) & 0x7Fu);
        } else if (cur == REF_LINEAR) {
            li = prim_base;
            lend = li + count;
        }
    };
    // the same for a reference out of a node or off the stack, which is never REF_LINEAR
    auto open_leaf_ref = [&]() {
        if ((cur & WIDE_LEAF) && cur < REF_POP) {
            li = prim_base + (cur & 0xFFFFFFu);
            lend = li + ((cur >> 24) & 0x7Fu);
        }
    };
    // Starts the search of one buffer.  Inside an instance an empty or missed buffer means what a
    // pop down to the instance's stack bottom means.
    auto begin_buffer = [&](const DevBuffer &buf) {
        node_base = buf.wide_first;
        prim_base = buf.prim_first;
        cur = walk_begin(buf, r);
        open_leaf(buf.prim_count);
        if (INSTANCES && cur == REF_NONE && inst_code != 0u) cur = REF_RETURN;
    };
#ifdef PT_WALK_STATS
    uint32_t ws_pass = 0, ws_busy = 0, ws_a = 0, ws_bpass = 0, ws_b = 0, ws_p = 0, ws_round = 0, ws_leaf = 0, ws_tailpass = 0, ws_tailbusy = 0;
#endif
    for (;;) {
#ifdef PT_WALK_STATS
        ws_round++;
#endif
#ifdef PT_BOUNDS_CHECK
        if (++rounds > WALK_MAX_PASSES) {
            if (lane_id() == 0u && atomicAdd(&bad_refs[1], 1ull) == 0ull) bad_refs[2] = 99ull;
            break;
        }
#endif
        // ---- deal new rays to idle lanes
        for (;;) {
            const unsigned long long idle = __ballot(phase == PHASE_IDLE);
            if (idle == 0ull) break;
            if (res_next == res_end) {
                if (used_up) break;
                uint32_t block;
                if (!blocks.next(block)) {
                    used_up = true;
                    break;
                }
                const uint32_t got = block * claim;
                res_next = got * 64u;
                res_end = (got + claim < n_pieces ? got + claim : n_pieces) * 64u;
            }
            const uint32_t want = (uint32_t)__popcll(idle), avail = res_end - res_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (phase == PHASE_IDLE && rank < avail && source(res_next + rank, slot, r)) {
                h.code = PTD_HIT_MISS;
                h.u = h.v = 0.0f;
                h.prim = 0u;
                phase = 0u;
                inst_code = 0u;
                lane_plain = ray_is_plain(r);
#ifdef PT_BOUNDS_CHECK
                passes = 0u;
#endif
                begin_buffer(set.tris);
            }
            res_next += want < avail ? want : avail;
        }
        if (__ballot(phase < PHASE_DONE) == 0ull) break;  // nothing in flight, so nothing left to claim either

        // ---- traverse until too few lanes are busy
        const bool can_refill = !used_up || res_next != res_end;
        // min/max form of the box test when no ray of the wave can produce a NaN in it
        bool plain = sc.boxes_finite != 0u && __ballot(phase < PHASE_DONE && !lane_plain) == 0ull;
        while (phase < PHASE_DONE) {
#ifdef PT_WALK_STATS
            {
                const uint32_t busy = (uint32_t)__popcll(__ballot(phase < PHASE_DONE));
                ws_pass++;
                ws_busy += busy;
                ws_a += (uint32_t)__popcll(__ballot(!(cur & WIDE_LEAF)));
                ws_leaf += (uint32_t)__popcll(__ballot(li < lend));
                ws_p += (uint32_t)__popcll(__ballot(cur == REF_POP));
                if (!can_refill) {
                    ws_tailpass++;
                    ws_tailbusy += busy;
                }
            }
#endif
#ifdef PT_BOUNDS_CHECK
            if (++passes > WALK_MAX_PASSES) {
                atomicAdd(&bad_refs[1], 1ull);
                WALK_CHECK(false, 5)
            }
#endif
            // (A) an interior node, for the lanes that have one pending; PT_WALK_NODE_STEPS times in a row, for the lanes
            // that are at a node again, before the wave looks at leaves, stacks and its own bookkeeping (two in a row:
            // +2.3 % on a 184 k-triangle scene, +3.2 % on 1 M triangles, +1.2 % on the interior; three: less than two;
            // unrolled, so that the second step is scheduled with the first: another +0.6 %)
#ifndef PT_WALK_NODE_STEPS
#define PT_WALK_NODE_STEPS 2
#endif
#pragma unroll
            for (int rep = 0; rep < PT_WALK_NODE_STEPS; rep++)
            if (!(cur & WIDE_LEAF)) {
                WALK_CHECK(node_base + cur < sc.n_wide, 1)
                const float4 *w = sc.wide + 4 * (size_t)(node_base + cur);
#if PT_TOP_LDS > 0
                float4 a, b, c, d;
                if (node_base == top_base && cur < top_n) {
                    a = g_top_nodes[4 * cur], b = g_top_nodes[4 * cur + 1], c = g_top_nodes[4 * cur + 2], d = g_top_nodes[4 * cur + 3];
                } else {
                    a = w[0], b = w[1], c = w[2], d = w[3];
                }
#else
                const float4 a = w[0], b = w[1], c = w[2], d = w[3];
#endif
                float el, er;
                bool hl, hr;
                // (one copy of this step per form of the test, chosen for the wave outside the per-lane condition, and the
                // same for the whole loop, were both measured 4 to 5 % slower)
                if (plain) {
                    slab_test_two_plain(a, b, c, r, hl, el, hr, er);
                } else {
                    hl = slab_test_entry(a.x, a.y, b.x, a.z, a.w, b.y, r, el);
                    hr = slab_test_entry(c.x, c.y, b.z, c.z, c.w, b.w, r, er);
                }
                const uint32_t lref = __float_as_uint(d.x), rref = __float_as_uint(d.y), axis = __float_as_uint(d.z);
                // one child passed: that one; none: the stack; both: the near one, which the ray's direction along the
                // node's axis decides (src/bvh.rs:757-771), with the far one stacked.  Written so that the ray's sign is
                // only looked at where both passed: selecting between the two outcomes per lane costs vector instructions.
                cur = hl ? lref : rref;
                if (!(hl || hr)) cur = REF_POP;
                if (hl && hr) {
                    const bool neg = ((r.neg >> axis) & 1u) != 0u;
                    st.push_flat(neg ? lref : rref, neg ? el : er);
                    cur = neg ? rref : lref;
                }
                open_leaf_ref();
            }
            // (B) one primitive of a pending leaf, once enough lanes have one or no lane has a node left
            {
                const unsigned long long at_leaf = __ballot(li < lend), at_node = __ballot(!(cur & WIDE_LEAF));
                if (at_leaf != 0ull && (at_node == 0ull || __popcll(at_leaf) >= leaf_batch)) {
#ifdef PT_WALK_STATS
                    ws_bpass++;
                    ws_b += (uint32_t)__popcll(at_leaf);
#endif
                    if (li < lend) {
                        if (INSTANCES && phase == PHASE_INSTS) {
                            // Instance::hit: into the instance's frame, then its set's three buffers
                            WALK_CHECK(li < sc.n_insts, 2)
                            const uint32_t gi = li++;
                            const float4 *ip = sc.insts + 7 * (size_t)gi;
                            const Affine inv = load_affine(ip);
                            const float4 last = ip[6];
                            child = __float_as_uint(last.x);
                            WALK_CHECK(child < sc.n_sets, 3)
                            WALK_CHECK((uint64_t)__float_as_uint(last.y) + __float_as_uint(last.z) <= sc.n_insts, 3)
                            st.push(cur, __uint_as_float(li));  // where to go on in the instance buffer
                            st.push(REF_RETURN, 0.0f);
                            const float t_min = r.t_min, t_max = r.t_max;
                            f3 lo = xf_point(inv, r.o), ld = xf_vector(inv, r.d);
                            chain_into(sc, last, lo, ld);  // an instance that was nested: the levels further in
                            r = make_trav_ray(lo, ld, t_min, t_max);
                            lane_plain = ray_is_plain(r);
                            inst_code = (gi - set.insts.prim_first + 1u) << 2;
                            phase = PHASE_CHILD;
                            li = lend = 0u;  // the leaf of the instance buffer is set aside (its place is in the record)
                            begin_buffer(sc.sets[child].tris);
                        } else {
                            const uint32_t type = phase & 3u;
                            WALK_CHECK(li < (type == 0u ? sc.n_tris : (type == 1u ? sc.n_spheres : sc.n_quads)), 4)
                            if (type == 0u) hit_tri_flat(sc, li, r, h, inst_code, tc);
                            else if (type == 1u) hit_prim<PT_PRIM_SPHERE>(sc, li, r, h, inst_code, tc);
                            else hit_quad_flat(sc, li, r, h, inst_code, tc);
                            li++;
                            if (li == lend) cur = REF_POP;
                            if (ANY_HIT && h.code != PTD_HIT_MISS) {  // nothing else about this ray matters
                                phase = PHASE_DONE;
                                cur = REF_NONE;
                                li = lend = 0u;
                                st.clear();
                            }
                        }
                    }
                }
            }
            // (P) the one place where the stack is popped
            if (cur == REF_POP) {
                cur = walk_pop<INSTANCES>(st, r);
                open_leaf_ref();
            }
            // (C) nothing left in this buffer: on to the next one, or out of the instance
            if (INSTANCES && cur == REF_RETURN) {
                WALK_CHECK(inst_code != 0u && phase >= PHASE_CHILD, 8)
                if (phase < PHASE_CHILD + 2u) {
                    phase++;
                    const DevSet &cs = sc.sets[child];
                    begin_buffer(phase == PHASE_CHILD + 1u ? cs.spheres : cs.quads);
                } else {
                    // back to the world frame and to the leaf of the instance buffer the lane came from
                    uint32_t leaf_ref = REF_NONE;
                    float next = 0.0f;
                    st.pop(0, leaf_ref, next);  // the REF_RETURN record
                    WALK_CHECK(leaf_ref == REF_RETURN, 9)
                    leaf_ref = REF_NONE;
                    st.pop(0, leaf_ref, next);  // (leaf of the instance buffer, next instance)
                    WALK_CHECK((leaf_ref & WIDE_LEAF) && leaf_ref != REF_NONE && leaf_ref != REF_RETURN, 6)
                    f3 o, d;
                    reload(slot, o, d);
                    const float t_min = r.t_min, t_max = r.t_max;
                    r = make_trav_ray(o, d, t_min, t_max);
                    lane_plain = ray_is_plain(r);
                    inst_code = 0u;
                    phase = PHASE_INSTS;
                    node_base = set.insts.wide_first;
                    prim_base = set.insts.prim_first;
                    cur = leaf_ref;
                    open_leaf(set.insts.prim_count);
                    li = __float_as_uint(next);
                    if (li >= lend) cur = REF_POP;
                }
            } else if (cur == REF_NONE && phase < PHASE_DONE) {
                // world level (inside an instance the bottom of the stack is REF_RETURN)
                WALK_CHECK(inst_code == 0u && st.sp == 0 && !st.has_top(), 7)
                phase++;
                if constexpr (tail_after) phase = PHASE_DONE;
                else if (phase == 1u) begin_buffer(set.spheres);
                else if (phase == 2u) begin_buffer(set.quads);
                else if (INSTANCES && phase == PHASE_INSTS) begin_buffer(set.insts);
                else phase = PHASE_DONE;
            }
            // lanes may have changed frame in this pass
            if (INSTANCES) plain = sc.boxes_finite != 0u && __ballot(phase < PHASE_DONE && !lane_plain) == 0ull;
            if (can_refill && __popcll(__ballot(phase < PHASE_DONE)) < refill_below) break;
        }

        // ---- finished rays leave
        if constexpr (tail_after) {
            const bool fin = phase == PHASE_DONE;
            auto leaf_of = [&](const DevBuffer &b, auto &&test) {
                if (b.prim_count == 0u) return;
                bool go = fin;
                if (b.tree != 0u) {  // walk_begin: the root's box
                    float e;
                    go = go && slab_test_entry(b.rmin[0], b.rmin[1], b.rmin[2], b.rmax[0], b.rmax[1], b.rmax[2], r, e);
                }
                const uint32_t first = b.prim_first + (b.tree != 0u ? (b.root_ref & 0xFFFFFFu) : 0u), count = b.tree != 0u ? ((b.root_ref >> 24) & 0x7Fu) : b.prim_count;
#pragma unroll 1
                for (uint32_t i = 0; i < count; i++)
                    if (go && !(ANY_HIT && h.code != PTD_HIT_MISS)) test(first + i);
            };
            leaf_of(set.spheres, [&](uint32_t g) { hit_prim<PT_PRIM_SPHERE>(sc, g, r, h, 0u, tc); });
            leaf_of(set.quads, [&](uint32_t g) { hit_quad_flat(sc, g, r, h, 0u, tc); });
        }
        sink(phase == PHASE_DONE, slot, r, h);
        if (phase == PHASE_DONE) phase = PHASE_IDLE;
    }
#ifdef PT_WALK_STATS
    if (lane_id() == 0u) {
        const uint32_t v[10] = {ws_pass, ws_busy, ws_a, ws_bpass, ws_b, ws_p, ws_round, ws_leaf, ws_tailpass, ws_tailbusy};
        for (int k = 0; k < 10; k++) atomicAdd(&g_walk_stats[k], (unsigned long long)v[k]);
    }
#endif
}

}  // namespace ptd
