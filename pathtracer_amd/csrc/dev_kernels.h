// The kernels of libptcore (launched from ptcore.hip): ray generation, the closest-hit kernels
// and the classification of finished rays into per-material queues, one shading kernel per
// material type, the shadow-ray kernel of the next-event estimator, accumulation, tone mapping
// and the first-hit query.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dev_packet.h"
#include "dev_shade.h"
#include "dev_walk.h"

using namespace ptd;

struct DevStats {
    unsigned long long rays_total, paths_total, extend_nonempty;
    unsigned long long node_tests, tri_tests, sphere_tests, quad_tests;
    // PT_BOUNDS_CHECK build only: [0] out-of-range references caught, [1] lanes that gave up after too
    // many passes, [2..7] state of the first such lane
    unsigned long long bad_refs[8];
    unsigned long long queue_overflow;  // queue entries dropped for want of room (wave_push)
    unsigned long long shadow_rays;     // of rays_total: shadow rays of the next-event estimator
};

struct BatchParams {
    uint64_t seed_mixed;
    uint32_t stratum_first;  // first stratum of this batch
    uint32_t n_strata;       // strata in this batch
    uint32_t npix;
    uint32_t n_slots;        // npix * n_strata
    float bsdf_prob;
    f3 background;
};
// Paths of a batch are numbered pixel-major: slot = first slot of the batch + pixel * n_strata + (stratum -
// stratum_first).  The 64 lanes of a wave then hold the strata of a few neighbouring pixels — almost the same
// ray — and a contiguous range of slots is a band of the image with all its strata.  The slot arrays are cut
// into regions of ps.region_cap slots, one batch per region (a single region where batches do not share
// launches), and a path's batch is found from its slot.
__device__ __forceinline__ void path_sample(const PathState &ps, uint32_t slot, uint64_t &seed_mixed, uint32_t &pixel, uint32_t &stratum) {
    const uint32_t region = slot / ps.region_cap;
    const RegionParams rp = ps.regions[region];
    const uint32_t local = slot - region * ps.region_cap;
    pixel = local / rp.n_strata;
    stratum = rp.stratum_first + (local - pixel * rp.n_strata);
    seed_mixed = rp.seed_mixed;
}

// Open chunks of the material queues, one set per wave of the workgroup, kept in LDS
// because their number is only known at run time.  Only the owning wave touches its row.
__shared__ uint32_t g_shade_push[BLOCK / 64][MAX_QUEUES][2];
__device__ __forceinline__ void shade_push_init() {
    const uint32_t l = lane_id();
    if (l < MAX_QUEUES) {
        g_shade_push[threadIdx.x / 64u][l][0] = QUEUE_EMPTY;
        g_shade_push[threadIdx.x / 64u][l][1] = 0u;
    }
}
__device__ __forceinline__ void shade_push_close(const PathState &ps) {
    for (uint32_t q = 0; q < ps.n_queues; q++) {
        WavePush w;
        w.chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_shade_push[threadIdx.x / 64u][q][0]);
        w.fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_shade_push[threadIdx.x / 64u][q][1]);
        wave_push_close(w, ps.shade_queue[q]);
    }
}

// Camera::get_ray (src/camera.rs:238-260) for one (pixel, stratum).
// `known` (0: unknown): the number of draws this ray is known to consume.  The generator is counter-based, so the
// draws that made the ray (the accepted point of the disk, then the two of the square) can be taken directly
// instead of repeating the rejection loop that led to them.
__device__ __forceinline__ void camera_ray(const DevCamera &cam, Rng &rng, uint32_t pixel, uint32_t stratum, f3 &o, f3 &d, uint32_t known = 0u) {
    const float u = (float)(pixel % cam.width), v = (float)(pixel / cam.width);
    const float s_u = (float)(stratum / cam.sqrt_spp), s_v = (float)(stratum % cam.sqrt_spp);
    float disk_u, disk_v;
    if (known >= 4u) {
        rng.counter += known - 4u;
        disk_u = rng.range(-1.0f, 1.0f);
        disk_v = rng.range(-1.0f, 1.0f);
    } else {
        for (;;) {  // DiskSampler::unit(), src/types/sampler.rs:49-58: drawn even when there is no lens
            float x = rng.range(-1.0f, 1.0f);
            float y = rng.range(-1.0f, 1.0f);
            if (x * x + y * y < 1.0f) {
                disk_u = x;
                disk_v = y;
                break;
            }
        }
    }
    f3 ray_center = cam.center;
    if (!(cam.defocus_angle <= 0.0f)) ray_center = cam.center + disk_u * cam.disk_u + disk_v * cam.disk_v;
    const float sx = rng.range(0.0f, 1.0f);  // SquareSampler((0.5, 0.5), 0.5)
    const float sy = rng.range(0.0f, 1.0f);
    const float offset_u = (s_u + sx) * cam.inv_sqrt_spp - 0.5f;
    const float offset_v = (s_v + sy) * cam.inv_sqrt_spp - 0.5f;
    const f3 pixel_center = cam.pixel00 + (u + offset_u) * cam.pixel_du + (v + offset_v) * cam.pixel_dv;
    o = ray_center;
    d = pixel_center - ray_center;
}

// A step's work list: the rays of the ray queue of its parity (`queued` positions, padding included), then the
// paths that start with the step ("fresh" paths: the `n_new` paths in slots new_base ..).  A fresh path is on its
// camera ray, which is never stored: whoever needs it computes it again from the path's pixel and stratum,
// together with what else a path carries at that point (draws consumed so far, full depth, throughput 1).
struct FreshPaths {
    uint32_t new_base, n_new;
};
struct WorkList {
    uint32_t parity, queued;
    FreshPaths fp;
    __device__ __forceinline__ bool fresh(uint32_t pos) const { return pos >= queued; }
    __device__ __forceinline__ uint32_t n_pieces() const { return queued / 64u + (fp.n_new + 63u) / 64u; }
    // the path at a position; QUEUE_EMPTY for padding and past the end
    __device__ __forceinline__ uint32_t slot(const PathState &ps, uint32_t pos) const {
        if (pos < queued) return ps.ray_queue[parity][pos];
        return pos - queued < fp.n_new ? fp.new_base + (pos - queued) : QUEUE_EMPTY;
    }
};
// The work list of a launch that carries no fresh paths (they have a kernel of their own beside it): the same, with
// "no position is fresh" known at compile time, so that the code for camera rays is not part of that launch.
struct QueuedOnly : WorkList {
    __device__ __forceinline__ bool fresh(uint32_t) const { return false; }
    __device__ __forceinline__ uint32_t slot(const PathState &ps, uint32_t pos) const { return pos < queued ? ps.ray_queue[parity][pos] : QUEUE_EMPTY; }
};
__device__ __forceinline__ WorkList work_list(const PathState &ps, uint32_t bounce, FreshPaths fp) {
    const uint32_t parity = bounce & 1u;
    const uint32_t n_chunks = min(ps.ctl[(CTL_RAY_CHUNKS + parity) * CTL_STRIDE], ps.max_ray_chunks);
    return WorkList{parity, n_chunks * QUEUE_CHUNK, fp};
}
__device__ __forceinline__ void fresh_path(const DevCamera &cam, const PathState &ps, uint32_t slot, f3 &o, f3 &d, uint32_t &draws, float &time,
                                           uint32_t known = 0u) {
    uint64_t seed_mixed;
    uint32_t pixel, stratum;
    path_sample(ps, slot, seed_mixed, pixel, stratum);
    Rng rng;
    rng.init(seed_mixed, pixel, stratum, 0u);
    camera_ray(cam, rng, pixel, stratum, o, d, known);
    draws = rng.counter;
    time = ps.moving ? path_time(rng.key) : 0.0f;
}
// Origin and direction of the ray at a position of the work list (which holds a path: not padding).
template <class WL>
__device__ __forceinline__ void path_ray(const DevCamera &cam, const PathState &ps, const WL &wl, uint32_t pos, f3 &o, f3 &d) {
    if (wl.fresh(pos)) {
        uint32_t draws;
        float time;
        fresh_path(cam, ps, wl.fp.new_base + (pos - wl.queued), o, d, draws, time);
        ps.fresh_draws[pos - wl.queued] = draws < 256u ? (uint8_t)draws : (uint8_t)0u;  // for the shading kernel
    } else {
        const float4 ro = ps.ray_o[wl.parity][pos], rd = ps.ray_d[wl.parity][pos];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
    }
}
// First launch of a step: the batch that starts with it (if any) gets its region, and the count of the rays the
// step's shading queues for the next step starts at zero.
__global__ void k_tick_begin(RegionParams *regions, uint32_t region, RegionParams rp, uint32_t set_region, uint32_t *next_rays) {
    if (set_region) regions[region] = rp;
    if (next_rays) *next_rays = 0u;
}

// Rays only (pt_generate_rays).
__global__ __launch_bounds__(BLOCK) void k_raygen_only(DevCamera cam, BatchParams bp, float *rays) {
    for (uint32_t pixel = blockIdx.x * BLOCK + threadIdx.x; pixel < bp.npix; pixel += gridDim.x * BLOCK) {
        Rng rng;
        rng.init(bp.seed_mixed, pixel, bp.stratum_first, 0u);
        f3 o, d;
        camera_ray(cam, rng, pixel, bp.stratum_first, o, d);
        float *r = rays + 8 * (size_t)pixel;
        r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = 0.001f; r[7] = INFINITY;
    }
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_down((int)v, o, 64);
    return v;
}

// After the closest hit is known: missed paths and paths that hit a primitive without
// material finish here (src/camera.rs:371, :277-283); the others are appended (by their position in the work
// list) to the queue of their material type.  Every lane of the wave must call this (inactive lanes with
// active == false), after shade_push_init() and before shade_push_close().
template <class WL>
__device__ __forceinline__ void classify(const DevScene &sc, const PathState &ps, const WL &wl, f3 background, bool active, uint32_t pos, uint32_t code,
                                         uint32_t prim) {
    uint32_t qslot = 0xFFu;
    if (active) {
        int32_t mat = -1;
        if (code != PTD_HIT_MISS) {
            const uint32_t type = code & 3u;
            mat = (type == PT_PRIM_TRI) ? sc.tri_mat[prim] : ((type == PT_PRIM_SPHERE) ? sc.sphere_mat[prim] : sc.quad_mat[prim]);
        }
        // without a surface to shade the path ends here, unless the scene has participating media: then it
        // may still scatter on its way out, and shade_vertex<PTD_MAT_MEDIUM> decides (medium_queue is 0xFF otherwise)
        qslot = mat >= 0 ? sc.mat_queue[mat] : sc.medium_queue;
        if (qslot == 0xFFu) {
            const uint32_t slot = wl.slot(ps, pos);
            const float4 b = wl.fresh(pos) ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : ps.beta[wl.parity][pos];
            if (!ps.nee) {
                const f3 l = (code == PTD_HIT_MISS) ? background : mk3(0.0f, 0.0f, 0.0f);
                ps.contrib[slot] = make_float4(b.x * l.x, b.y * l.y, b.z * l.z, 0.0f);
            } else if (code == PTD_HIT_MISS) {  // contrib already holds the light samples of the earlier vertices
                const float4 c = ps.contrib[slot];
                ps.contrib[slot] = make_float4(c.x + b.x * background.x, c.y + b.y * background.y, c.z + b.z * background.z, 0.0f);
            }
        }
    }
    // one pass per material queue that some lane of this wave needs
    unsigned long long todo = __ballot(qslot != 0xFFu);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)qslot, leader);
        const bool mine = qslot == q;
        uint32_t *state = g_shade_push[threadIdx.x / 64u][q];
        WavePush w;
        w.chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)state[0]);
        w.fill = (uint32_t)__builtin_amdgcn_readfirstlane((int)state[1]);
        wave_push(w, mine, pos, ps.shade_queue[q], &ps.ctl[(CTL_SHADE_CHUNKS + 2u * q + wl.parity) * CTL_STRIDE], ps.max_chunks, ps.queue_overflow);
        if (lane_id() == 0u) {
            state[0] = w.chunk;
            state[1] = w.fill;
        }
        todo &= ~__ballot(mine);
    }
}

// What every closest-hit kernel does first: clear the control words the shade kernels of
// this bounce will fill, and count the launch.
__device__ __forceinline__ void extend_prologue(const PathState &ps, uint32_t bounce, uint32_t n_chunks, uint32_t n_new, DevStats *stats) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t parity = bounce & 1u;
        ps.ctl[(CTL_RAY_CHUNKS + (parity ^ 1u)) * CTL_STRIDE] = 0u;
        for (uint32_t l = 0; l < WORK_LISTS; l++) {
            ps.ctl[(CTL_PACKET_CURSOR + WORK_LISTS * (parity ^ 1u) + l) * CTL_STRIDE] = 0u;
            ps.ctl[(CTL_WALK_CURSOR + WORK_LISTS * (parity ^ 1u) + l) * CTL_STRIDE] = 0u;
        }
        if (n_chunks + n_new) atomicAdd(&stats->extend_nonempty, 1ull);
        if (n_new) {  // the paths that start with this launch, and their camera rays
            atomicAdd(&stats->paths_total, (unsigned long long)n_new);
            atomicAdd(&stats->rays_total, (unsigned long long)n_new);
        }
    }
}

__device__ __forceinline__ void flush_counters(bool count, const TravCounters &tc, DevStats *stats) {
    if (!count) return;
    const uint32_t a = wave_sum(tc.nodes), b = wave_sum(tc.tris), c = wave_sum(tc.spheres), d = wave_sum(tc.quads);
    if (lane_id() == 0) {
        if (a) atomicAdd(&stats->node_tests, (unsigned long long)a);
        if (b) atomicAdd(&stats->tri_tests, (unsigned long long)b);
        if (c) atomicAdd(&stats->sphere_tests, (unsigned long long)c);
        if (d) atomicAdd(&stats->quad_tests, (unsigned long long)d);
    }
}

// Closest hit among the world's triangles, spheres and quads for every entry of the step's work list (first
// three lines of PrimitiveBuffer::hit, src/objects.rs:192-194).
// CLASSIFY: the scene has no instances, so this is the whole of PrimitiveBuffer::hit.
// A scene with moving spheres (pts_moving_sphere) tests them last, where they are at the path's time.
// The time of the path named by `slot`: a hash of its generator's key.
__device__ __forceinline__ float path_time_of_slot(const PathState &ps, uint32_t slot) {
    uint64_t seed_mixed;
    uint32_t pixel, stratum;
    path_sample(ps, slot, seed_mixed, pixel, stratum);
    Rng rng;
    rng.init(seed_mixed, pixel, stratum, 0u);
    return path_time(rng.key);
}
template <class WL>
__device__ __forceinline__ float path_time_at(const DevCamera &cam, const PathState &ps, const WL &wl, uint32_t pos) {
    // a queued ray carries its path's time in beta.w, unless that word holds a density (next-event estimation)
    if (!wl.fresh(pos)) return ps.nee ? path_time_of_slot(ps, wl.slot(ps, pos)) : ps.beta[wl.parity][pos].w;
    f3 o, d;
    uint32_t draws;
    float time;
    fresh_path(cam, ps, wl.fp.new_base + (pos - wl.queued), o, d, draws, time);
    return time;
}
__device__ __forceinline__ void hit_moving_spheres(const DevScene &sc, float time, TravRay &r, HitState &h, TravCounters &tc) {
    for (uint32_t k = 0; k < sc.n_moving; k++) {
        const uint32_t g = sc.n_static_spheres + k;
        const float4 sp = sc.spheres[g], dl = sc.moving_delta[k];
        const f3 c = mk3(sp.x, sp.y, sp.z) + time * mk3(dl.x, dl.y, dl.z);
        tc.spheres++;
        hit_sphere(make_float4(c.x, c.y, c.z, sp.w), g, r, h, 0u);
    }
}
template <bool COUNT, bool CLASSIFY>
__global__ __launch_bounds__(BLOCK) void k_extend(DevScene sc, DevCamera cam, PathState ps, uint32_t bounce, f3 background, FreshPaths fp, DevStats *stats) {
    const WorkList wl = work_list(ps, bounce, fp);
    extend_prologue(ps, bounce, wl.queued, fp.n_new, stats);
    if (CLASSIFY) shade_push_init();
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    TravCounters tc{0, 0, 0, 0};
    const uint32_t n_pieces = wl.n_pieces();
    for (uint32_t piece = wave_index(); piece < n_pieces; piece += wave_count()) {
        const uint32_t pos = piece * 64u + lane_id();
        const bool active = wl.slot(ps, pos) != QUEUE_EMPTY;
        HitState h;
        h.code = PTD_HIT_MISS;
        h.prim = 0;
        if (active) {
            f3 o, d;
            path_ray(cam, ps, wl, pos, o, d);
            TravRay r = make_trav_ray(o, d, 0.001f, INFINITY);
            hit_world_prims(sc, r, h, st, tc);
            if (CLASSIFY && sc.n_moving != 0u) hit_moving_spheres(sc, path_time_at(cam, ps, wl, pos), r, h, tc);
            ps.hit[pos] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
            ps.hit2[pos] = h.code;
        }
        if (CLASSIFY) classify(sc, ps, wl, background, active, pos, h.code, h.prim);
    }
    if (CLASSIFY) shade_push_close(ps);
    flush_counters(COUNT, tc, stats);
}

// The whole of PrimitiveBuffer::hit with rays drawn by the waves as their lanes fall idle
// (dev_walk.h); this is the form used for rendering whenever the scene's leaves fit the
// 64-byte node layout.  INSTANCES: the world has an instance buffer.
// Five workgroups of four waves per CU is what the LDS stacks allow; the register budget follows from it
// (at most 96 VGPRs per lane for five waves per SIMD).
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 5
#endif
#define WALK_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PT_WALK_WAVES, PT_WALK_WAVES)))
// MOVING: the scene has moving spheres (pts_moving_sphere): a finished ray is tested against them, where
// they are at its path's time, before it is classified.
// FRESH: the launch may carry paths that start with it (false: they are in k_extend_packet's hands, fp.n_new is 0).
// TAIL: walk_tail_applies(sc.world), checked by the host: the world's few spheres and quads are tested when a ray is handed on.
template <bool INSTANCES, bool MOVING, bool FRESH, bool TAIL = false>
__global__ __launch_bounds__(BLOCK) WALK_OCCUPANCY void k_extend_walk(DevScene sc, DevCamera cam, PathState ps, uint32_t bounce, f3 background, FreshPaths fp,
                                                                      DevStats *stats) {
    const WorkList full = work_list(ps, bounce, fp);
    typedef typename std::conditional<FRESH, WorkList, QueuedOnly>::type List;
    const List wl{full};
    extend_prologue(ps, bounce, wl.queued, fp.n_new, stats);
    const uint32_t n_pieces = wl.n_pieces();
    if (n_pieces == 0u) return;
    shade_push_init();
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    // a lane's ray is named by its position in the work list
    walk_world<INSTANCES, false, TAIL>(
        sc, n_pieces, &ps.ctl[(CTL_WALK_CURSOR + WORK_LISTS * wl.parity) * CTL_STRIDE], ps.refill_below, ps.leaf_batch, st, stats->bad_refs,
        [&](uint32_t i, uint32_t &pos, TravRay &r) {
            if (wl.slot(ps, i) == QUEUE_EMPTY) return false;
            pos = i;
            f3 o, d;
            path_ray(cam, ps, wl, pos, o, d);
            r = make_trav_ray(o, d, 0.001f, INFINITY);
            return true;
        },
        [&](uint32_t pos, f3 &o, f3 &d) { path_ray(cam, ps, wl, pos, o, d); },
        [&](bool done, uint32_t pos, const TravRay &r, const HitState &h) {
            if constexpr (MOVING) {
                TravRay r2 = r;
                HitState h2 = h;
                if (done) {
                    TravCounters tc{0, 0, 0, 0};
                    hit_moving_spheres(sc, path_time_at(cam, ps, wl, pos), r2, h2, tc);
                    ps.hit[pos] = make_float4(r2.t_max, h2.u, h2.v, __uint_as_float(h2.prim));
                    ps.hit2[pos] = h2.code;
                }
                classify(sc, ps, wl, background, done, pos, h2.code, h2.prim);
            } else {
                if (done) {
                    ps.hit[pos] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
                    ps.hit2[pos] = h.code;
                }
                classify(sc, ps, wl, background, done, pos, h.code, h.prim);
            }
        });
    shade_push_close(ps);
}

// PrimitiveBuffer::hit for the paths that start with this step, 64 consecutive slots per wave: a few neighbouring
// pixels with all the strata of the batch, rays that differ by a fraction of a pixel.  The wave searches for
// all of them at once (dev_packet.h); each lane's own sequence of tests is the reference's.  For scenes whose
// BVHs all have reference-order nodes; the rays queued by earlier steps go through k_extend_walk, launched
// beside this kernel with no fresh paths of its own.
template <bool INSTANCES, bool MOVING>
__global__ __launch_bounds__(BLOCK) void k_extend_packet(DevScene sc, DevCamera cam, PathState ps, uint32_t bounce, f3 background, FreshPaths fp, DevStats *stats) {
    const WorkList wl = work_list(ps, bounce, fp);
    if (blockIdx.x == 0 && threadIdx.x == 0 && fp.n_new) {
        if (wl.queued == 0u) atomicAdd(&stats->extend_nonempty, 1ull);  // else k_extend_walk counts the step
        atomicAdd(&stats->paths_total, (unsigned long long)fp.n_new);
        atomicAdd(&stats->rays_total, (unsigned long long)fp.n_new);
    }
    shade_push_init();
    const uint32_t n_pieces = (fp.n_new + 63u) / 64u;
#ifndef PT_PACKET_CLAIM
#define PT_PACKET_CLAIM 2
#endif
    // Waves take the pieces as they get to them, in blocks of PT_PACKET_CLAIM (BlockClaim): pieces differ in cost,
    // sky against geometry, and the waves of this grid do not all start together beside the walking kernel.
    // Measured on a 184 k-triangle scene at 768x768 (closest-hit time of ten 256-strata steps): pieces dealt out in
    // advance, wave w taking w, w + waves, ...: 266 ms; one shared counter, blocks of 32 / 8 / 4 / 2 pieces: 327 / 266 /
    // 256 / 274 ms; a counter per XCD, blocks of 8 / 4 / 2 / 1: 264 / 253 / 249 / 250 ms; shading 97 -> 91 ms either way.
    BlockClaim claim;
    claim.init(&ps.ctl[(CTL_PACKET_CURSOR + WORK_LISTS * wl.parity) * CTL_STRIDE], (n_pieces + (uint32_t)PT_PACKET_CLAIM - 1u) / (uint32_t)PT_PACKET_CLAIM);
    uint32_t piece = 0u, piece_end = 0u;
    for (;; piece++) {
        if (piece == piece_end) {
            uint32_t block;
            if (!claim.next(block)) break;
            piece = block * (uint32_t)PT_PACKET_CLAIM;
            piece_end = min(piece + (uint32_t)PT_PACKET_CLAIM, n_pieces);
        }
        const uint32_t i = piece * 64u + lane_id();
        const bool active = i < fp.n_new;
        const uint32_t pos = wl.queued + i;
        HitState h;
        h.code = PTD_HIT_MISS;
        h.u = h.v = 0.0f;
        h.prim = 0u;
        f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(1.0f, 1.0f, 1.0f);
        float time = 0.0f;
        if (active) {
            uint32_t draws;
            fresh_path(cam, ps, fp.new_base + i, o, d, draws, time);
            ps.fresh_draws[i] = draws < 256u ? (uint8_t)draws : (uint8_t)0u;  // for the shading kernel
        }
        TravRay r = make_trav_ray(o, d, 0.001f, INFINITY);
        const bool plain = sc.boxes_finite != 0u && __ballot(active && !ray_is_plain(r)) == 0ull;
        packet_hit_world<INSTANCES>(sc, __ballot(active), plain, r, h);
        if (active) {
            if constexpr (MOVING) {
                TravCounters tc{0, 0, 0, 0};
                hit_moving_spheres(sc, time, r, h, tc);
            }
            ps.hit[pos] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
            ps.hit2[pos] = h.code;
        }
        classify(sc, ps, wl, background, active, pos, h.code, h.prim);
    }
    shade_push_close(ps);
}

// The instance buffer of the world (last line of PrimitiveBuffer::hit, src/objects.rs:195),
// continuing from the hit and t_max the first kernel left; then the moving spheres, if any, and classification.
template <bool COUNT>
__global__ __launch_bounds__(BLOCK) void k_extend_inst(DevScene sc, DevCamera cam, PathState ps, uint32_t bounce, f3 background, FreshPaths fp, DevStats *stats) {
    const WorkList wl = work_list(ps, bounce, fp);
    shade_push_init();
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    TravCounters tc{0, 0, 0, 0};
    const uint32_t n_pieces = wl.n_pieces();
    for (uint32_t piece = wave_index(); piece < n_pieces; piece += wave_count()) {
        const uint32_t pos = piece * 64u + lane_id();
        const bool active = wl.slot(ps, pos) != QUEUE_EMPTY;
        HitState h;
        h.code = PTD_HIT_MISS;
        h.prim = 0;
        if (active) {
            f3 o, d;
            path_ray(cam, ps, wl, pos, o, d);
            const float4 h4 = ps.hit[pos];
            h.u = h4.y;
            h.v = h4.z;
            h.prim = __float_as_uint(h4.w);
            h.code = ps.hit2[pos];
            TravRay r = make_trav_ray(o, d, 0.001f, h4.x);
            const uint32_t before = h.code;
            const float t_before = r.t_max;
            hit_world_insts(sc, r, h, st, tc);
            if (sc.n_moving != 0u) hit_moving_spheres(sc, path_time_at(cam, ps, wl, pos), r, h, tc);
            if (h.code != before || r.t_max != t_before || h.prim != __float_as_uint(h4.w)) {
                ps.hit[pos] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
                ps.hit2[pos] = h.code;
            }
        }
        classify(sc, ps, wl, background, active, pos, h.code, h.prim);
    }
    shade_push_close(ps);
    flush_counters(COUNT, tc, stats);
}

constexpr float SHADOW_REACH = 0.999f;  // a shadow ray ends this short of the light it is aimed at
// What a vertex leaves behind: the path's next ray with what travels with it (alive), and with next-event
// estimation possibly a shadow ray.
struct VertexOut {
    bool alive, shadow;
    float4 ray_o, ray_d, beta;            // as PathState stores them
    float4 shadow_o, shadow_d, shadow_c;  // origin, direction, factor on the emitter's radiance
};

// The surface part of shade_vertex below.
template <int MT, bool NEE>
__device__ __forceinline__ void shade_surface(const DevScene &sc, const PathState &ps, const BatchParams &bp, uint32_t slot, VertexOut &out,
                                              f3 ro, f3 rd, float4 h4, float4 b4, uint32_t code, f3 beta, uint32_t depth, uint32_t rng_counter) {
    const Surf s = make_surf(sc, ro, rd, h4.x, h4.y, h4.z, __float_as_uint(h4.w), code, NEE ? (ps.moving ? path_time_of_slot(ps, slot) : 0.0f) : b4.w);
    const DevMaterial m = sc.mats[s.material];
    if constexpr (MT == PTS_MAT_DIFFUSE_LIGHT) {
        // Diffuse::emitted (light.rs:42-47); an emissive hit ends the path (camera.rs:288-290)
        f3 e = mk3(0.0f, 0.0f, 0.0f);
        if (s.front_face) e = tex_value(sc, m.texture, s.u, s.v, s.p);
        if constexpr (NEE) {
            float w = 1.0f;
            // against the density with which light sampling yields this very point: the terms of the lights the ray meets here
            if (sc.n_light_total != 0u && b4.w > 0.0f && b4.w < INFINITY) w = b4.w / (b4.w + light_pdf<true, true>(sc, ro, rd, h4.x));
            const float4 c = ps.contrib[slot];
            const f3 add = cmul(beta, e) * w;
            ps.contrib[slot] = make_float4(c.x + add.x, c.y + add.y, c.z + add.z, 0.0f);
        } else {
            ps.contrib[slot] = make_float4(beta.x * e.x, beta.y * e.y, beta.z * e.z, 0.0f);
        }
    } else {
        uint64_t seed_mixed;
        uint32_t pixel, stratum;
        path_sample(ps, slot, seed_mixed, pixel, stratum);
        Rng rng;
        rng.init(seed_mixed, pixel, stratum, rng_counter);
        const Onb surface = make_onb(s.normal);
        const f3 w_out = -normalize(to_local(surface, rd));
        f3 w_in_world;
        bool go_on = true;
        float drawn_with = 0.0f;  // NEE: density of the ray that leaves this vertex, 0 = specular
        if constexpr (MT == PTS_MAT_METAL || MT == PTS_MAT_DIELECTRIC) {
            // purely specular: no pdf, no roulette (camera.rs:298-313)
            const f3 w_in = mat_scatter<MT>(sc, m, rng, w_out, s);
            const f3 attenuation = mat_eval<MT>(sc, m, w_out, w_in, s);
            w_in_world = to_world(surface, w_in);
            beta = cmul(beta, attenuation);
        } else if constexpr (NEE && (MT == PTS_MAT_LAMBERTIAN || MT == PTS_MAT_DISNEY_DIFFUSE || MT == PTS_MAT_DISNEY_SHEEN)) {
            if (sc.n_light_total != 0u) {
                // one light, a point on it; the sample counts with that light's own density over directions and its
                // radiance where the ray meets it, if the shadow ray, which ends just short of it, hits nothing
                uint32_t picked;
                const f3 ls = light_sample<true>(sc, rng, s.p, &picked);
                LightHit lh;
                const float lp = light_term<true>(sc, picked, s.p, ls, &lh) / (float)sc.n_light_total;
                if (lp > 0.0f && lp < INFINITY) {
                    const f3 wl = normalize(to_local(surface, ls));
                    const float sp = mat_pdf<MT>(m, w_out, wl, s);
                    if (sp > 0.0f) {
                        const f3 f = mat_eval<MT>(sc, m, w_out, wl, s) * fabsf(wl.y);
                        const f3 c = cmul(beta, f) / (lp + sp);
                        const int32_t lmat = lh.type == PT_PRIM_TRI ? sc.tri_mat[lh.prim] : (lh.type == PT_PRIM_SPHERE ? sc.sphere_mat[lh.prim] : sc.quad_mat[lh.prim]);
                        f3 le = mk3(0.0f, 0.0f, 0.0f);
                        if (lmat >= 0 && sc.mats[lmat].type == PTS_MAT_DIFFUSE_LIGHT) {  // Diffuse::emitted (light.rs:42-47)
                            const Surf sl = make_surf(sc, s.p, ls, lh.t, lh.u, lh.v, lh.prim, lh.type);
                            if (sl.front_face) le = tex_value(sc, sc.mats[lmat].texture, sl.u, sl.v, sl.p);
                        }
                        if (le.x != 0.0f || le.y != 0.0f || le.z != 0.0f) {
                            const f3 add = cmul(c, le);
                            out.shadow_o = make_float4(s.p.x, s.p.y, s.p.z, 0.0f);
                            out.shadow_d = make_float4(ls.x, ls.y, ls.z, lh.t * SHADOW_REACH);
                            out.shadow_c = make_float4(add.x, add.y, add.z, 0.0f);
                            out.shadow = true;
                        }
                    }
                }
            }
            const f3 w_in = mat_scatter<MT>(sc, m, rng, w_out, s);
            w_in_world = to_world(surface, w_in);
            const float sp = mat_pdf<MT>(m, w_out, w_in, s);
            const f3 bsdf = mat_eval<MT>(sc, m, w_out, w_in, s);
            const float cosv = fabsf(w_in.y);
            const f3 throughput = (bsdf * cosv) / sp;
            const float weight = 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
            if (weight != weight || rng.gen_bool(weight)) {
                go_on = false;
            } else {
                beta = cmul(beta, throughput / (1.0f - weight));
                drawn_with = sp;
            }
        } else {
            // mixture of BSDF sampling and light sampling (camera.rs:317-357)
            f3 w_in;
            bool is_bsdf;
            if (rng.gen_bool(bp.bsdf_prob)) {
                w_in = mat_scatter<MT>(sc, m, rng, w_out, s);
                is_bsdf = true;
                w_in_world = to_world(surface, w_in);
            } else {
                const f3 ls = light_sample(sc, rng, s.p);
                w_in = normalize(to_local(surface, ls));
                is_bsdf = false;
                w_in_world = ls;
            }
            if (!is_bsdf) depth = 2u;
            const float scatter_pdf = mat_pdf<MT>(m, w_out, w_in, s);
            // with probability 1 for BSDF sampling the light term is a product with zero: 0, or NaN
            const float l_pdf = bp.bsdf_prob == 1.0f ? light_pdf_times_zero(sc, s.p, w_in_world) : light_pdf(sc, s.p, w_in_world);
            const float pdf = bp.bsdf_prob * scatter_pdf + (1.0f - bp.bsdf_prob) * l_pdf;
            const f3 bsdf = mat_eval<MT>(sc, m, w_out, w_in, s);
            const float cosv = fabsf(w_in.y);
            const f3 throughput = (bsdf * cosv) / pdf;
            const float weight = !is_bsdf ? 0.0f : 1.0f - rclamp(rmax(throughput.z, rmax(throughput.y, throughput.x)), 0.0f, 1.0f);
            if (weight != weight || rng.gen_bool(weight)) {
                go_on = false;  // returns Color::gray(0): contrib stays zero
            } else {
                beta = cmul(beta, throughput / (1.0f - weight));
            }
        }
        if (go_on) {
            depth -= 1u;
            if (depth == 0u) {
                // the next ray_color call returns zeros without tracing (camera.rs:271-273)
                if constexpr (!NEE) ps.contrib[slot] = make_float4(beta.x * 0.0f, beta.y * 0.0f, beta.z * 0.0f, 0.0f);
            } else {
                out.ray_o = make_float4(s.p.x, s.p.y, s.p.z, __uint_as_float(rng.counter));
                out.ray_d = make_float4(w_in_world.x, w_in_world.y, w_in_world.z, __uint_as_float(depth));
                out.beta = make_float4(beta.x, beta.y, beta.z, NEE ? drawn_with : b4.w);  // w: the path's time unless NEE
                out.alive = true;
            }
        }
    }
}

// One path vertex: the body of Camera::ray_color after the hit (src/camera.rs:286-369), in
// multiplicative form, for the path `slot` at position `pos` of the step's work list, whose hit
// (ps.hit, ps.hit2) has a material of type MT.  out.alive: the path goes on with out.ray_o / ray_d / beta.
//
// NEE selects the integrator that is not in the reference (PT_OPT_INTEGRATOR = 1): next-event
// estimation with one light sample per vertex and the balance heuristic, at the vertices whose
// material has a scattering_pdf that is the density of its scatter (Lambertian, DisneyDiffuse,
// DisneySheen: checked numerically in tests/test_material_pdfs.py; for the other Disney lobes of
// the reference it is not, so no combination of the two samplers can reproduce what the
// reference's estimator converges to, and those vertices are handled exactly as the reference
// does, with BSDF sampling alone).  At a vertex with light sampling a direction is drawn with
// LightBuffer::sample; when its density lp is positive and finite and the BSDF sampler could
// have produced it (sp > 0), out.shadow is set and out.shadow_o / shadow_d / shadow_c hold the shadow ray and
// beta * f * |cos| / (lp + sp) for the shadow-ray kernel.  The path itself continues by
// BSDF sampling with the reference's roulette; an emitter found that way counts with weight
// sp' / (sp' + lp'), and fully after the camera or a vertex without light sampling.  Radiance is
// added to ps.contrib as it is found; paths that end without finding any add nothing.
template <int MT, bool NEE, bool MEDIA = false>
__device__ __forceinline__ void shade_vertex(const DevScene &sc, const DevCamera &cam, const PathState &ps, const BatchParams &bp, const WorkList &wl, uint32_t pos,
                                             uint32_t slot, VertexOut &out) {
    f3 ro, rd;
    float4 b4;
    uint32_t depth;        // the max_depth argument of this ray_color call, >= 1
    uint32_t rng_counter;  // draws the path has consumed so far
    out.alive = out.shadow = false;
    if (wl.fresh(pos)) {  // the camera ray is not stored anywhere: src/camera.rs:207-210 with the path's own generator again
        float time;
        fresh_path(cam, ps, slot, ro, rd, rng_counter, time, ps.fresh_draws[pos - wl.queued]);
        depth = cam.max_depth;
        b4 = make_float4(1.0f, 1.0f, 1.0f, NEE ? 0.0f : time);  // w: the path's time, or the density the ray was drawn with (none)
    } else {
        const float4 ro4 = ps.ray_o[wl.parity][pos], rd4 = ps.ray_d[wl.parity][pos];
        ro = mk3(ro4.x, ro4.y, ro4.z);
        rd = mk3(rd4.x, rd4.y, rd4.z);
        rng_counter = __float_as_uint(ro4.w);
        depth = __float_as_uint(rd4.w);
        b4 = ps.beta[wl.parity][pos];
    }
    const float4 h4 = ps.hit[pos];
    const uint32_t code = ps.hit2[pos];
    f3 beta = mk3(b4.x, b4.y, b4.z);
    if constexpr (MT == PTD_MAT_MEDIUM || MEDIA) {
        // participating media (not in the reference): the ray may scatter before it reaches what it hit
        uint64_t seed_mixed;
        uint32_t pixel, stratum;
        path_sample(ps, slot, seed_mixed, pixel, stratum);
        Rng rng;
        rng.init(seed_mixed, pixel, stratum, rng_counter);
        const bool missed = MT == PTD_MAT_MEDIUM && code == PTD_HIT_MISS;
        float t_event;
        uint32_t which = 0u;
        if (medium_event(sc, rng, ro, rd, missed ? INFINITY : h4.x, t_event, which)) {
            const f3 p = ro + t_event * rd;
            const f3 albedo = tex_value(sc, sc.media[which].texture, 0.0f, 0.0f, p);
            const f3 dir = medium_direction(rng);
            beta = cmul(beta, albedo);
            depth -= 1u;
            if (depth == 0u) {
                if constexpr (!NEE) ps.contrib[slot] = make_float4(beta.x * 0.0f, beta.y * 0.0f, beta.z * 0.0f, 0.0f);
                return;
            }
            out.ray_o = make_float4(p.x, p.y, p.z, __uint_as_float(rng.counter));
            out.ray_d = make_float4(dir.x, dir.y, dir.z, __uint_as_float(depth));
            out.beta = make_float4(beta.x, beta.y, beta.z, NEE ? 0.0f : b4.w);  // no light is sampled at a scattering event
            out.alive = true;
            return;
        }
        rng_counter = rng.counter;
        if constexpr (MT == PTD_MAT_MEDIUM) {  // nothing to shade: src/camera.rs:371, :277-283
            const f3 l = missed ? bp.background : mk3(0.0f, 0.0f, 0.0f);
            if constexpr (NEE) {  // contrib already holds the light samples of the earlier vertices
                if (missed) {
                    const float4 c = ps.contrib[slot];
                    ps.contrib[slot] = make_float4(c.x + beta.x * l.x, c.y + beta.y * l.y, c.z + beta.z * l.z, 0.0f);
                }
            } else {
                ps.contrib[slot] = make_float4(beta.x * l.x, beta.y * l.y, beta.z * l.z, 0.0f);
            }
            return;
        }
    }
    if constexpr (MT != PTD_MAT_MEDIUM) shade_surface<MT, NEE>(sc, ps, bp, slot, out, ro, rd, h4, b4, code, beta, depth, rng_counter);
}

// shade_vertex for every entry of the queue of material type MT; survivors go to the next ray queue, their state
// to the same position of the arrays that run beside it.
// The queue may mix batches (each at a different bounce): `bp` holds what they have in common (sampling
// probability, background), the rest comes from the path's region.
// MODE: 0 = the reference's estimator, 1 = next-event estimation, 2 = the reference's estimator in a scene with
// participating media, 3 = next-event estimation in such a scene.
template <int MT, int MODE>
__global__ __launch_bounds__(BLOCK) void k_shade(DevScene sc, DevCamera cam, PathState ps, BatchParams bp, uint32_t bounce, uint32_t q, FreshPaths fp, DevStats *stats) {
    constexpr bool NEE = MODE == 1 || MODE == 3, MEDIA = MODE == 2 || MODE == 3;
    // a path ends at an emitter (src/camera.rs:288-290), unless a medium scatters it on its way there
    constexpr bool CONTINUES = MT != PTS_MAT_DIFFUSE_LIGHT || MEDIA;
    const WorkList wl = work_list(ps, bounce, fp);
    const uint32_t parity = wl.parity;
    const uint32_t n_chunks = min(ps.ctl[(CTL_SHADE_CHUNKS + 2u * q + parity) * CTL_STRIDE], ps.max_chunks);
    if (blockIdx.x == 0 && threadIdx.x == 0) ps.ctl[(CTL_SHADE_CHUNKS + 2u * q + (parity ^ 1u)) * CTL_STRIDE] = 0u;
    const uint32_t *queue = ps.shade_queue[q];
    WavePush next, shadows;
    next.init();
    shadows.init();
    uint32_t survivors = 0u, shadow_rays = 0u;
    // The queue's pieces either dealt out in advance (ps.shade_claim == 0: wave w takes w, w + waves, ...) or drawn in
    // blocks of ps.shade_claim chunks as the waves get to them (BlockClaim).  The host chooses per step: where most
    // vertices are first hits, already queued in path order, dealing in advance is 9 % faster; where most belong to
    // later bounces, drawing is (8 % on 1 M triangles in a box).
    constexpr uint32_t CHUNK_PIECES = QUEUE_CHUNK / 64u;
    const uint32_t n_pieces = n_chunks * CHUNK_PIECES, claim_chunks = n_chunks >= ps.shade_claim_from ? ps.shade_claim : 0u;
    BlockClaim claim;
    claim.init(&ps.ctl[(CTL_SHADE_CURSOR + WORK_LISTS * (2u * q + parity)) * CTL_STRIDE], claim_chunks ? (n_chunks + claim_chunks - 1u) / claim_chunks : 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (uint32_t l = 0; l < WORK_LISTS; l++) ps.ctl[(CTL_SHADE_CURSOR + WORK_LISTS * (2u * q + (parity ^ 1u)) + l) * CTL_STRIDE] = 0u;
    uint32_t piece = wave_index(), piece_end = claim_chunks ? piece : n_pieces;
    const uint32_t piece_step = claim_chunks ? 1u : wave_count();
    for (;; piece += piece_step) {
        if (piece >= piece_end) {
            uint32_t block;
            if (claim_chunks == 0u || !claim.next(block)) break;
            piece = block * claim_chunks * CHUNK_PIECES;
            piece_end = min(piece + claim_chunks * CHUNK_PIECES, n_pieces);
        }
        const uint32_t pos = queue[(size_t)piece * 64u + lane_id()];
        VertexOut v;
        v.alive = v.shadow = false;
        uint32_t slot = QUEUE_EMPTY;
        if (pos != QUEUE_EMPTY) {
            slot = wl.slot(ps, pos);
            shade_vertex<MT, NEE, MEDIA>(sc, cam, ps, bp, wl, pos, slot, v);
        }
        if constexpr (CONTINUES) {
            survivors += (uint32_t)__popcll(__ballot(v.alive));
            const uint32_t to = wave_push(next, v.alive, slot, ps.ray_queue[parity ^ 1u], &ps.ctl[(CTL_RAY_CHUNKS + (parity ^ 1u)) * CTL_STRIDE], ps.max_ray_chunks,
                                          ps.queue_overflow);
            if (to != QUEUE_EMPTY) {
                ps.ray_o[parity ^ 1u][to] = v.ray_o;
                ps.ray_d[parity ^ 1u][to] = v.ray_d;
                ps.beta[parity ^ 1u][to] = v.beta;
            }
        }
        if constexpr (NEE) {
            shadow_rays += (uint32_t)__popcll(__ballot(v.shadow));
            const uint32_t to = wave_push(shadows, v.shadow, slot, ps.shadow_queue, &ps.ctl[(CTL_SHADOW_CHUNKS + parity) * CTL_STRIDE], ps.max_ray_chunks, ps.queue_overflow);
            if (to != QUEUE_EMPTY) {
                ps.shadow_o[to] = v.shadow_o;
                ps.shadow_d[to] = v.shadow_d;
                ps.shadow_c[to] = v.shadow_c;
            }
        }
    }
    if constexpr (CONTINUES) wave_push_close(next, ps.ray_queue[parity ^ 1u]);
    if constexpr (NEE) wave_push_close(shadows, ps.shadow_queue);
    if ((survivors | shadow_rays) && lane_id() == 0u) {
        atomicAdd(&stats->rays_total, (unsigned long long)(survivors + shadow_rays));
        if (shadow_rays) atomicAdd(&stats->shadow_rays, (unsigned long long)shadow_rays);
        if (survivors && ps.next_rays) atomicAdd(ps.next_rays, survivors);
        if (survivors && ps.bounce_rays && !NEE && bounce + 1u < BOUNCE_PROFILE_LEN) atomicAdd(&ps.bounce_rays[bounce + 1u], survivors);  // read back to size later launches
    }
}

// Shadow rays of one step (next-event estimation): each ends just short of the light it is aimed at, and the
// first hit along it blocks it; an unblocked one adds what its vertex left behind (throughput x BSDF x cosine x
// emitted radiance over the sum of the two densities) to its path, dimmed by the media it crosses.
template <bool INSTANCES>
__global__ __launch_bounds__(BLOCK) WALK_OCCUPANCY void k_shadow_walk(DevScene sc, PathState ps, uint32_t bounce, DevStats *stats) {
    const uint32_t parity = bounce & 1u;
    const uint32_t n_chunks = min(ps.ctl[(CTL_SHADOW_CHUNKS + parity) * CTL_STRIDE], ps.max_ray_chunks);
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // for the shading of the next step
        ps.ctl[(CTL_SHADOW_CHUNKS + (parity ^ 1u)) * CTL_STRIDE] = 0u;
        for (uint32_t l = 0; l < WORK_LISTS; l++) ps.ctl[(CTL_SHADOW_WALK_CURSOR + WORK_LISTS * (parity ^ 1u) + l) * CTL_STRIDE] = 0u;
    }
    if (n_chunks == 0u) return;
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    const uint32_t *queue = ps.shadow_queue;
    // a lane's ray is named by its position in the shadow queue
    walk_world<INSTANCES, true>(
        sc, n_chunks * (QUEUE_CHUNK / 64u), &ps.ctl[(CTL_SHADOW_WALK_CURSOR + WORK_LISTS * parity) * CTL_STRIDE], ps.refill_below, ps.leaf_batch, st, stats->bad_refs,
        [&](uint32_t i, uint32_t &pos, TravRay &r) {
            if (queue[i] == QUEUE_EMPTY) return false;
            pos = i;
            const float4 ro = ps.shadow_o[i], rd = ps.shadow_d[i];
            r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.001f, rd.w);  // rd.w: where the ray ends
            return true;
        },
        [&](uint32_t pos, f3 &o, f3 &d) {
            const float4 ro = ps.shadow_o[pos], rd = ps.shadow_d[pos];
            o = mk3(ro.x, ro.y, ro.z);
            d = mk3(rd.x, rd.y, rd.z);
        },
        [&](bool done, uint32_t pos, const TravRay &r_in, const HitState &h_in) {
            if (!done || h_in.code != PTD_HIT_MISS) return;
            const float4 ro = ps.shadow_o[pos], rd = ps.shadow_d[pos];
            const uint32_t slot = queue[pos];
            if (sc.n_moving != 0u) {  // the moving spheres, where they are at the path's time
                TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.001f, rd.w);
                HitState h = h_in;
                TravCounters tc{0, 0, 0, 0};
                hit_moving_spheres(sc, path_time_of_slot(ps, slot), r, h, tc);
                if (h.code != PTD_HIT_MISS) return;
            }
            const float4 f = ps.shadow_c[pos], c = ps.contrib[slot];
            f3 add = mk3(f.x, f.y, f.z);
            if (sc.n_media != 0u) add = add * medium_transmittance(sc, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), rd.w);
            ps.contrib[slot] = make_float4(c.x + add.x, c.y + add.y, c.z + add.z, 0.0f);
        });
}

// Paths of one batch that are still alive when the batch is due: each lane takes such a path out of the ray
// queue the step's shading has just filled and alternates closest hit (the reference-order loop) and
// shade_vertex until the path ends.  This happens under light sampling only: a vertex that samples a light
// resets the depth left to 2 (src/camera.rs:332), so a path that keeps doing that outlives max_depth, which
// is all the schedules plan for.  `region`: the batch's region of the slot arrays.  The path's state stays
// where its ray's queue entry was (position `pos` of the arrays of the next step's parity, and of the hit
// arrays, which the finished step no longer needs); the queue entry becomes padding.
// The scan comes first and on its own (k_finish_scan: a few registers, every wave of the chip; the path-finishing
// code needs 256 and runs one wave per SIMD, which made its own scan of 14 M queue entries cost 220 us a step): it
// lists (position, slot) of the region's entries in ps.finish_list and turns them into padding.  k_finish<MEDIA,
// true> then finishes the listed paths; should more be left than the list holds (it holds 2^20), the rest stay in
// the queue and k_finish<MEDIA, false>, which returns at once otherwise, scans for them as before.
__global__ __launch_bounds__(BLOCK) void k_finish_scan(PathState ps, uint32_t bounce, uint32_t region, uint32_t k) {
    const uint32_t np = (bounce & 1u) ^ 1u;
    const uint32_t n_chunks = min(ps.ctl[(CTL_RAY_CHUNKS + np) * CTL_STRIDE], ps.max_ray_chunks);
    uint32_t *count = &ps.ctl[(CTL_FINISH_COUNT + k) * CTL_STRIDE];
    if (blockIdx.x == 0 && threadIdx.x == 0) ps.ctl[(CTL_FINISH_COUNT + (k ^ 1u)) * CTL_STRIDE] = 0u;  // for the scan after this one
    uint4 *queue4 = (uint4 *)ps.ray_queue[np];
    uint32_t *queue = ps.ray_queue[np];
    const uint32_t lo = region * ps.region_cap, hi = lo + ps.region_cap;  // below 2^31 (slots are)
    const uint32_t n4 = n_chunks * (QUEUE_CHUNK / 4u);
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n4; i += gridDim.x * BLOCK) {
        const uint4 q = queue4[i];
        const uint32_t s[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (uint32_t j = 0; j < 4u; j++) {
            if (s[j] < lo || s[j] >= hi) continue;  // QUEUE_EMPTY is above every slot
            const uint32_t at = atomicAdd(count, 1u);
            if (at < ps.finish_cap) {
                ps.finish_list[at] = make_uint2(4u * i + j, s[j]);
                queue[4u * i + j] = QUEUE_EMPTY;
            }
        }
    }
}

// One leftover path to its end: the reference-order closest-hit loop and shade_vertex in turn.
template <bool MEDIA>
__device__ __forceinline__ void finish_path(const DevScene &sc, const DevCamera &cam, const PathState &ps, const BatchParams &bp, uint32_t np, const WorkList &wl,
                                            uint32_t pos, uint32_t slot, f3 background, bool insts, LaneStack &st, TravCounters &tc, uint32_t &traced) {
    for (bool first = true;; first = false) {
        if (!first) traced++;
        const float4 ro = ps.ray_o[np][pos], rd = ps.ray_d[np][pos];
        TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), 0.001f, INFINITY);
        HitState h;
        hit_world_prims(sc, r, h, st, tc);
        if (insts) hit_world_insts(sc, r, h, st, tc);
        if (sc.n_moving != 0u) hit_moving_spheres(sc, ps.beta[np][pos].w, r, h, tc);
        int32_t mat = -1;
        if (h.code != PTD_HIT_MISS) {
            const uint32_t type = h.code & 3u;
            mat = (type == PT_PRIM_TRI) ? sc.tri_mat[h.prim] : ((type == PT_PRIM_SPHERE) ? sc.sphere_mat[h.prim] : sc.quad_mat[h.prim]);
        }
        ps.hit[pos] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
        ps.hit2[pos] = h.code;
        VertexOut v;
        v.alive = v.shadow = false;
        if (mat < 0) {
            if constexpr (MEDIA) {
                shade_vertex<PTD_MAT_MEDIUM, false, true>(sc, cam, ps, bp, wl, pos, slot, v);
            } else {  // src/camera.rs:371, :277-283
                const float4 b = ps.beta[np][pos];
                const f3 l = (h.code == PTD_HIT_MISS) ? background : mk3(0.0f, 0.0f, 0.0f);
                ps.contrib[slot] = make_float4(b.x * l.x, b.y * l.y, b.z * l.z, 0.0f);
            }
        } else {
            switch (sc.mats[mat].type) {
#define FINISH(MT) case MT: shade_vertex<MT, false, MEDIA>(sc, cam, ps, bp, wl, pos, slot, v); break;
                FINISH(PTS_MAT_LAMBERTIAN)
                FINISH(PTS_MAT_METAL)
                FINISH(PTS_MAT_DIELECTRIC)
                FINISH(PTS_MAT_DIFFUSE_LIGHT)
                FINISH(PTS_MAT_DISNEY)
                FINISH(PTS_MAT_DISNEY_DIFFUSE)
                FINISH(PTS_MAT_DISNEY_METAL)
                FINISH(PTS_MAT_DISNEY_GLASS)
                FINISH(PTS_MAT_DISNEY_CLEARCOAT)
                FINISH(PTS_MAT_DISNEY_SHEEN)
#undef FINISH
                default: break;
            }
        }
        if (!v.alive) break;
        ps.ray_o[np][pos] = v.ray_o;
        ps.ray_d[np][pos] = v.ray_d;
        ps.beta[np][pos] = v.beta;
    }
}

// LISTED: the paths k_finish_scan listed; otherwise: a scan of the queue, only where the list was too short.
template <bool MEDIA, bool LISTED>
__global__ __launch_bounds__(BLOCK) void k_finish(DevScene sc, DevCamera cam, PathState ps, BatchParams bp, uint32_t bounce, uint32_t region, f3 background,
                                                  uint32_t k, DevStats *stats) {
    const uint32_t found = ps.ctl[(CTL_FINISH_COUNT + k) * CTL_STRIDE];
    if (!LISTED && found <= ps.finish_cap) return;
    const uint32_t np = (bounce & 1u) ^ 1u;
    const WorkList wl{np, 0xFFFFFFFFu, FreshPaths{0u, 0u}};  // every position is a queue position
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    TravCounters tc{0, 0, 0, 0};
    const bool insts = sc.world.insts.prim_count != 0u;
    uint32_t traced = 0u;  // rays beyond the queued ones (those are already counted)
    if constexpr (LISTED) {
        const uint32_t n = min(found, ps.finish_cap);
        for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
            const uint2 e = ps.finish_list[i];
            finish_path<MEDIA>(sc, cam, ps, bp, np, wl, e.x, e.y, background, insts, st, tc, traced);
        }
    } else {
        const uint32_t n_chunks = min(ps.ctl[(CTL_RAY_CHUNKS + np) * CTL_STRIDE], ps.max_ray_chunks);
        uint32_t *queue = ps.ray_queue[np];
        const uint32_t n_pos = n_chunks * QUEUE_CHUNK;
        for (uint32_t pos = blockIdx.x * BLOCK + threadIdx.x; pos < n_pos; pos += gridDim.x * BLOCK) {
            const uint32_t slot = queue[pos];
            if (slot == QUEUE_EMPTY || slot / ps.region_cap != region) continue;
            queue[pos] = QUEUE_EMPTY;
            finish_path<MEDIA>(sc, cam, ps, bp, np, wl, pos, slot, background, insts, st, tc, traced);
        }
    }
    if (traced) atomicAdd(&stats->rays_total, (unsigned long long)traced);
}

// `*pixel += pixel_color` once per stratum, in stratum order, NaN samples skipped (camera.rs:212-217).  The
// radiance slots are left at zero for the batch that uses them next: a path that ends without finding light
// (Russian roulette, src/camera.rs:355-357) writes nothing.
// Each workgroup takes BLOCK consecutive pixels at a time: their slots are one contiguous stretch, which the
// workgroup copies to LDS with coalesced loads (and clears with coalesced stores); each thread then adds up its
// own pixel's strata, in order, from LDS.
constexpr uint32_t ACC_LDS_SLOTS = 2560;  // 40 KB
__global__ __launch_bounds__(BLOCK) void k_accumulate(float4 *contrib, float *accum, uint32_t npix, uint32_t n_strata) {
    __shared__ float4 tile[ACC_LDS_SLOTS];
    const uint32_t per_pass = ACC_LDS_SLOTS / n_strata < (uint32_t)BLOCK ? ACC_LDS_SLOTS / n_strata : (uint32_t)BLOCK;  // pixels per pass
    if (per_pass == 0u) {  // more strata per pixel than the tile holds: straight from memory
        for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < npix; p += gridDim.x * BLOCK) {
            float r = accum[3 * p], g = accum[3 * p + 1], b = accum[3 * p + 2];
            for (uint32_t k = 0; k < n_strata; k++) {
                const float4 c = contrib[(size_t)p * n_strata + k];
                contrib[(size_t)p * n_strata + k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (c.x != c.x || c.y != c.y || c.z != c.z) continue;
                r += c.x;
                g += c.y;
                b += c.z;
            }
            accum[3 * p] = r;
            accum[3 * p + 1] = g;
            accum[3 * p + 2] = b;
        }
        return;
    }
    for (uint32_t first = blockIdx.x * per_pass; first < npix; first += gridDim.x * per_pass) {
        const uint32_t n_pix = npix - first < per_pass ? npix - first : per_pass;
        const uint32_t n_slots = n_pix * n_strata;
        float4 *src = contrib + (size_t)first * n_strata;
        for (uint32_t i = threadIdx.x; i < n_slots; i += BLOCK) {
            tile[i] = src[i];
            src[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();
        if (threadIdx.x < n_pix) {
            const uint32_t p = first + threadIdx.x;
            float r = accum[3 * p], g = accum[3 * p + 1], b = accum[3 * p + 2];
            for (uint32_t k = 0; k < n_strata; k++) {
                const float4 c = tile[threadIdx.x * n_strata + k];
                if (c.x != c.x || c.y != c.y || c.z != c.z) continue;
                r += c.x;
                g += c.y;
                b += c.z;
            }
            accum[3 * p] = r;
            accum[3 * p + 1] = g;
            accum[3 * p + 2] = b;
        }
        __syncthreads();
    }
}

// ColorOps::to_rgb (src/types/color.rs:36-42, :68-78) of sum / samples (camera.rs:223-225).
__device__ __forceinline__ float linear_to_gamma(float c) {
    if (c < 0.0031308f) return 12.92f * c;
    return 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}
__global__ __launch_bounds__(BLOCK) void k_tonemap(const float *accum, uint8_t *rgb, uint32_t npix, float samples) {
    for (uint32_t p = blockIdx.x * BLOCK + threadIdx.x; p < npix; p += gridDim.x * BLOCK) {
        for (int c = 0; c < 3; c++) {
            const float lin = accum[3 * p + c] / samples;
            uint32_t v = f2u32_sat(255.999f * linear_to_gamma(lin));
            rgb[3 * p + c] = (uint8_t)(v > 255u ? 255u : v);
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_load_rays(const float *rays, PathState ps, uint32_t n) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float *r = rays + 8 * (size_t)i;
        ps.ray_o[0][i] = make_float4(r[0], r[1], r[2], r[6]);
        ps.ray_d[0][i] = make_float4(r[3], r[4], r[5], r[7]);
        ps.ray_queue[0][i] = i;
    }
    const uint32_t n_chunks = (n + QUEUE_CHUNK - 1u) / QUEUE_CHUNK;
    if (blockIdx.x == 0) {
        for (uint32_t i = n + threadIdx.x; i < n_chunks * QUEUE_CHUNK; i += BLOCK) ps.ray_queue[0][i] = QUEUE_EMPTY;
        if (threadIdx.x == 0) {
            ps.ctl[(CTL_RAY_CHUNKS + 0) * CTL_STRIDE] = n_chunks;
            for (uint32_t l = 0; l < WORK_LISTS; l++) ps.ctl[(CTL_WALK_CURSOR + l) * CTL_STRIDE] = 0u;
        }
    }
}

// Closest hit for caller-supplied rays with their own [t_min, t_max]: the same two parts.
__global__ __launch_bounds__(BLOCK) void k_trace(DevScene sc, PathState ps, uint32_t n) {
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    TravCounters tc{0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 ro = ps.ray_o[0][i], rd = ps.ray_d[0][i];
        TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
        HitState h;
        hit_world_prims(sc, r, h, st, tc);
        ps.hit[i] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
        ps.hit2[i] = h.code;
    }
}
// The same for 64 consecutive caller-supplied rays per wave, searched together (dev_packet.h): whatever the rays
// have in common or not, each gets the hit it gets alone.
template <bool INSTANCES>
__global__ __launch_bounds__(BLOCK) void k_trace_packet(DevScene sc, PathState ps, uint32_t n) {
    const uint32_t n_pieces = (n + 63u) / 64u;
    for (uint32_t piece = wave_index(); piece < n_pieces; piece += wave_count()) {
        const uint32_t i = piece * 64u + lane_id();
        const bool active = i < n;
        float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        if (active) ro = ps.ray_o[0][i], rd = ps.ray_d[0][i];
        TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
        HitState h;
        h.code = PTD_HIT_MISS;
        h.u = h.v = 0.0f;
        h.prim = 0u;
        const bool plain = sc.boxes_finite != 0u && __ballot(active && !ray_is_plain(r)) == 0ull;
        packet_hit_world<INSTANCES>(sc, __ballot(active), plain, r, h);
        if (active) {
            ps.hit[i] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
            ps.hit2[i] = h.code;
        }
    }
}
template <bool INSTANCES, bool TAIL = false>
__global__ __launch_bounds__(BLOCK) void k_trace_walk(DevScene sc, PathState ps, uint32_t n, DevStats *stats) {
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    walk_world<INSTANCES, false, TAIL>(
        sc, (n + QUEUE_CHUNK - 1u) / QUEUE_CHUNK * (QUEUE_CHUNK / 64u), &ps.ctl[(CTL_WALK_CURSOR + 0) * CTL_STRIDE], ps.refill_below, ps.leaf_batch, st, stats->bad_refs,
        [&](uint32_t i, uint32_t &slot, TravRay &r) {
            slot = ps.ray_queue[0][i];
            if (slot == QUEUE_EMPTY) return false;
            const float4 ro = ps.ray_o[0][slot], rd = ps.ray_d[0][slot];
            r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, rd.w);
            return true;
        },
        [&](uint32_t slot, f3 &o, f3 &d) {
            const float4 ro = ps.ray_o[0][slot], rd = ps.ray_d[0][slot];
            o = mk3(ro.x, ro.y, ro.z);
            d = mk3(rd.x, rd.y, rd.z);
        },
        [&](bool done, uint32_t slot, const TravRay &r, const HitState &h) {
            if (done) {
                ps.hit[slot] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
                ps.hit2[slot] = h.code;
            }
        });
}
__global__ __launch_bounds__(BLOCK) void k_trace_inst(DevScene sc, PathState ps, uint32_t n) {
    LaneStack st;
    st.init(ps.overflow, ps.overflow_stride);
    TravCounters tc{0, 0, 0, 0};
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 ro = ps.ray_o[0][i], rd = ps.ray_d[0][i], h4 = ps.hit[i];
        HitState h;
        h.u = h4.y;
        h.v = h4.z;
        h.prim = __float_as_uint(h4.w);
        h.code = ps.hit2[i];
        TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, h4.x);
        hit_world_insts(sc, r, h, st, tc);
        ps.hit[i] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
        ps.hit2[i] = h.code;
    }
}
// The moving spheres, where they are at times[i] (null: time 0), against ray i with the hit and t_max the static
// scene left: the last line of the renderer's closest-hit step, for caller-supplied rays.
__global__ __launch_bounds__(BLOCK) void k_trace_moving(DevScene sc, PathState ps, uint32_t n, const float *times) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 ro = ps.ray_o[0][i], rd = ps.ray_d[0][i], h4 = ps.hit[i];
        HitState h;
        h.u = h4.y;
        h.v = h4.z;
        h.prim = __float_as_uint(h4.w);
        h.code = ps.hit2[i];
        TravRay r = make_trav_ray(mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ro.w, h4.x);
        const float time = times ? times[i] : 0.0f;
        for (uint32_t k = 0; k < sc.n_moving; k++) {
            const uint32_t g = sc.n_static_spheres + k;
            const float4 sp = sc.spheres[g], dl = sc.moving_delta[k];
            const f3 c = mk3(sp.x, sp.y, sp.z) + time * mk3(dl.x, dl.y, dl.z);
            hit_sphere(make_float4(c.x, c.y, c.z, sp.w), g, r, h, 0u);
        }
        ps.hit[i] = make_float4(r.t_max, h.u, h.v, __uint_as_float(h.prim));
        ps.hit2[i] = h.code;
    }
}
__global__ __launch_bounds__(BLOCK) void k_trace_out(DevScene sc, PathState ps, uint32_t n, pt_hit *out) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float4 h4 = ps.hit[i];
        const uint32_t code = ps.hit2[i], prim = __float_as_uint(h4.w);
        pt_hit o;
        o.t = h4.x;
        o.u = h4.y;
        o.v = h4.z;
        if (code == PTD_HIT_MISS) {
            o.prim_type = -1;
            o.prim_id = -1;
            o.inst_id = -1;
            o.u = o.v = 0.0f;
        } else {
            const uint32_t type = code & 3u, inst1 = code >> 2;
            o.prim_type = (int32_t)type;
            o.inst_id = (int32_t)inst1 - 1;
            const DevSet *set = &sc.sets[0];
            if (inst1) set = &sc.sets[__float_as_uint(sc.insts[7 * (size_t)(sc.world.insts.prim_first + inst1 - 1u) + 6].x)];
            const uint32_t first = (type == PT_PRIM_TRI) ? set->tris.prim_first : ((type == PT_PRIM_SPHERE) ? set->spheres.prim_first : set->quads.prim_first);
            o.prim_id = (int32_t)(prim - first);
            if (type == PT_PRIM_SPHERE && sc.n_moving != 0u && prim >= sc.n_static_spheres) {  // belongs to no set
                o.prim_type = PT_PRIM_MOVING_SPHERE;
                o.prim_id = (int32_t)(prim - sc.n_static_spheres);
            }
        }
        out[i] = o;
    }
}

// The Material trait of one material at caller-supplied shading points (pt_material_probe): the functions
// the shading kernels call, one probe per lane.
template <int MT>
__device__ __forceinline__ void probe_material(const DevScene &sc, const DevMaterial &m, uint64_t seed_mixed, const pt_material_probe_in &in,
                                               pt_material_probe_out &out) {
    Surf s;
    s.p = mk3(in.p[0], in.p[1], in.p[2]);
    s.normal = mk3(0.0f, 1.0f, 0.0f);
    s.t = 1.0f;
    s.front_face = in.front_face != 0u;
    s.material = 0;
    s.u = in.u;
    s.v = in.v;
    const f3 w_out = mk3(in.w_out[0], in.w_out[1], in.w_out[2]), w_at = mk3(in.w_in[0], in.w_in[1], in.w_in[2]);
    Rng rng;
    rng.init(seed_mixed, in.pixel, in.stratum, 0u);
    f3 w_in = mk3(0.0f, 0.0f, 0.0f), e = mk3(0.0f, 0.0f, 0.0f);
    if constexpr (MT == PTS_MAT_DIFFUSE_LIGHT) {
        if (s.front_face) e = tex_value(sc, m.texture, s.u, s.v, s.p);  // Diffuse::emitted, light.rs:42-47
    } else {
        w_in = mat_scatter<MT>(sc, m, rng, w_out, s);
    }
    const f3 f = mat_eval<MT>(sc, m, w_out, w_in, s), f_at = mat_eval<MT>(sc, m, w_out, w_at, s);
    out.w_in[0] = w_in.x; out.w_in[1] = w_in.y; out.w_in[2] = w_in.z;
    out.draws = rng.counter;
    out.bsdf[0] = f.x; out.bsdf[1] = f.y; out.bsdf[2] = f.z;
    out.pdf = mat_pdf<MT>(m, w_out, w_in, s);
    out.bsdf_at[0] = f_at.x; out.bsdf_at[1] = f_at.y; out.bsdf_at[2] = f_at.z;
    out.pdf_at = mat_pdf<MT>(m, w_out, w_at, s);
    out.emitted[0] = e.x; out.emitted[1] = e.y; out.emitted[2] = e.z;
    out.reserved = 0u;
}
__global__ __launch_bounds__(BLOCK) void k_material_probe(DevScene sc, uint32_t material, uint64_t seed_mixed, const pt_material_probe_in *in, uint32_t n,
                                                          pt_material_probe_out *out) {
    const DevMaterial m = sc.mats[material];
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const pt_material_probe_in pi = in[i];
        pt_material_probe_out po;
        switch (m.type) {
            case PTS_MAT_LAMBERTIAN: probe_material<PTS_MAT_LAMBERTIAN>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_METAL: probe_material<PTS_MAT_METAL>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DIELECTRIC: probe_material<PTS_MAT_DIELECTRIC>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DIFFUSE_LIGHT: probe_material<PTS_MAT_DIFFUSE_LIGHT>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DISNEY: probe_material<PTS_MAT_DISNEY>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DISNEY_DIFFUSE: probe_material<PTS_MAT_DISNEY_DIFFUSE>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DISNEY_METAL: probe_material<PTS_MAT_DISNEY_METAL>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DISNEY_GLASS: probe_material<PTS_MAT_DISNEY_GLASS>(sc, m, seed_mixed, pi, po); break;
            case PTS_MAT_DISNEY_CLEARCOAT: probe_material<PTS_MAT_DISNEY_CLEARCOAT>(sc, m, seed_mixed, pi, po); break;
            default: probe_material<PTS_MAT_DISNEY_SHEEN>(sc, m, seed_mixed, pi, po); break;
        }
        out[i] = po;
    }
}

// LightBuffer::pdf for caller-supplied rays (origin and direction of rays[i][8]).
__global__ __launch_bounds__(BLOCK) void k_light_pdf(DevScene sc, const float *rays, uint32_t n, int tris_by_solid_angle, float *out) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const float *r = rays + 8 * (size_t)i;
        const f3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]);
        out[i] = tris_by_solid_angle ? light_pdf<true>(sc, o, d) : light_pdf<false>(sc, o, d);
    }
}
