// Producer side of the chunked work queues (layout: dev_scene.h).
#pragma once
#include "dev_scene.h"

namespace ptd {

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// A wave's open chunk of one queue.  Both members are the same on all lanes.
struct WavePush {
    uint32_t chunk, fill;
    __device__ __forceinline__ void init() {
        chunk = QUEUE_EMPTY;
        fill = 0u;
    }
};

// Pads the rest of the open chunk; call once per wave and queue when the wave is done with it.
__device__ __forceinline__ void wave_push_close(const WavePush &w, uint32_t *entries) {
    if (w.chunk == QUEUE_EMPTY) return;
    for (uint32_t i = w.fill + lane_id(); i < QUEUE_CHUNK; i += 64u) entries[(size_t)w.chunk * QUEUE_CHUNK + i] = QUEUE_EMPTY;
}

// Appends `id` for every lane with want == true.  All 64 lanes of the wave must call it
// together.  `n_chunks` is the queue's chunk count.  The host sizes every queue for the worst
// case (paths / (QUEUE_CHUNK - 63) + one open chunk per producing wave of every launch that
// fills it); should a queue still run out of its `max_chunks` chunks, the push is dropped and
// counted in *overflowed, which the next synchronising call reports as an error, instead of
// writing past the end of the array.
// Returns the position of the lane's entry in the queue (QUEUE_EMPTY for lanes that push nothing, or whose push
// was dropped): arrays that run parallel to a queue are indexed by it.
__device__ __forceinline__ uint32_t wave_push(WavePush &w, bool want, uint32_t id, uint32_t *entries, uint32_t *n_chunks, uint32_t max_chunks,
                                              unsigned long long *overflowed) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return QUEUE_EMPTY;
    const uint32_t k = (uint32_t)__popcll(mask);
    if (w.chunk == QUEUE_EMPTY || w.fill + k > QUEUE_CHUNK) {
        wave_push_close(w, entries);
        uint32_t got = 0u;
        if (lane_id() == 0u) got = atomicAdd(n_chunks, 1u);
        w.chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        w.fill = 0u;
        if (w.chunk >= max_chunks) {
            // consumers walk min(chunk count, max_chunks) chunks, so the count may stay above the capacity
            if (lane_id() == 0u) atomicAdd(overflowed, (unsigned long long)k);
            w.chunk = QUEUE_EMPTY;
            return QUEUE_EMPTY;
        }
    }
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t pos = w.chunk * QUEUE_CHUNK + w.fill + rank;
    if (want) entries[pos] = id;
    w.fill += k;
    return want ? pos : QUEUE_EMPTY;
}

// Work handed out as the waves get to it.  Block b of the work is on list b mod WORK_LISTS; a wave draws from the
// list of its XCD (a counter on a line of its own) and, when that is used up, from the next ones.  All XCDs
// sweep the work together from its start to its end, whatever the number of workgroups a CU holds at once and
// whenever they start: what is read and written at any one time lies close together, and results are appended
// to the queues nearly in the order of the work.  `cursors`: WORK_LISTS control words, CTL_STRIDE apart, zero
// before the launch.
struct BlockClaim {
    uint32_t *cursors;
    uint32_t n_blocks, list, lists_left;
    __device__ __forceinline__ void init(uint32_t *c, uint32_t n) {
        cursors = c;
        n_blocks = n;
        list = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) % WORK_LISTS;  // HW_REG_XCC_ID[3:0]
        lists_left = WORK_LISTS;
    }
    // the next block for this wave (the same value in every lane); false when the work is used up
    __device__ __forceinline__ bool next(uint32_t &block) {
        while (lists_left != 0u) {
            uint32_t got = 0u;
            if (lane_id() == 0u) got = atomicAdd(&cursors[list * CTL_STRIDE], 1u);
            block = (uint32_t)__builtin_amdgcn_readfirstlane((int)got) * WORK_LISTS + list;
            if (block < n_blocks) return true;
            lists_left--;
            list = (list + 1u) % WORK_LISTS;
        }
        return false;
    }
};

// Identity of the calling wave in the launch, and the number of waves in it.
__device__ __forceinline__ uint32_t wave_index() { return blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; }
__device__ __forceinline__ uint32_t wave_count() { return gridDim.x * (blockDim.x / 64u); }

}  // namespace ptd
