// libptcore: the MI355X (gfx950) path-tracing core behind include/ptcore.h — the host side:
// context, options, scene upload, path state, the two schedules and the entry points.  The
// kernels are in dev_kernels.h.
//
// Wavefront organisation of Camera::render (src/camera.rs:162-236) and the recursion of
// Camera::ray_color (src/camera.rs:262-373), unrolled: a batch holds every pixel of k
// consecutive strata as independent paths; a step is one closest-hit launch, which also sorts
// the paths by material type into per-type queues, and one shading launch per material type
// present in the scene, which evaluates the vertex, applies Russian roulette and appends the
// survivors to the next ray queue.  In the rolling schedule a step carries bounce 0 of a new
// batch together with the later bounces of the batches before it; the other schedule runs the
// bounces of one batch after another.  Work sizes are read from device memory, so steps follow
// each other without host synchronisation.  Finished paths leave their radiance in a per-path
// slot and a final kernel adds the slots of each pixel in stratum order, so the accumulator is
// deterministic and sums in the order the reference does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <deque>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "dev_kernels.h"
#include "ptcore_internal.h"

// ============================================================================ host side

struct pt_context {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    std::string err;

    // options
    float bsdf_prob = 1.0f;
    uint64_t max_paths = 32000000;
    bool profile = false, counters = false;
    bool force_ref_order = false;  // PT_OPT_TRAVERSAL = 1
    bool trace_packet = false;     // PT_OPT_TRAVERSAL = 2: first-hit queries a wave at a time where the scene allows it
    bool wide_ok = false;          // every leaf fits the packed wide-node reference
    uint32_t stack_need = 0;       // bound on traversal stack entries for this scene
    bool use_wide() const { return wide_ok && !force_ref_order; }
    // The paths that start with a step are searched a wave at a time (k_extend_packet): scenes whose BVHs all come
    // with reference-order nodes (packet_ok, decided at upload).
    bool packet_ok = false, packet = true;  // PT_OPT_PACKET
    bool use_packet() const { return packet && packet_ok && use_wide() && !counters; }
#ifdef PT_WALK_WAVES
    int blocks_per_cu = PT_WALK_WAVES;
#else
    int blocks_per_cu = 5;  // as many as the LDS stacks allow: measured 0.4 - 4 % faster than 4 on every scene
#endif
    int refill_below = 32, leaf_batch = 16, shade_blocks_per_cu = 8;
    int shade_claim = -1;  // PT_OPT_SHADE_CLAIM
    bool nee = false;  // PT_OPT_INTEGRATOR = 1: next-event estimation (shade_vertex)
    uint32_t device_bvh_min = 0;  // triangle buffers that come without nodes and have at least this many triangles get their BVH built on the device (0: never)
    double live_factor = 4.0;  // PT_OPT_LIVE_FACTOR
    // PT_OPT_BATCH_TICKS: launches a batch of the rolling schedule takes part in before what is left of it is
    // finished in one go (k_finish); its slots are held that long.  0: max_depth launches.
    uint32_t batch_ticks = 0;  // 0: learnt from the scene (auto_ticks, once a drain has been seen), 1: max_depth
    uint32_t auto_ticks = 0;   // 0: not known yet
    uint32_t roll_regions() const {
        if (nee) return cam.max_depth;  // k_finish knows the reference's estimator only
        const uint32_t want = batch_ticks >= 2u ? batch_ticks : (batch_ticks == 0u && auto_ticks != 0u ? auto_ticks : cam.max_depth);
        return want < cam.max_depth ? want : cam.max_depth;
    }
    bool shade_spread = true;  // the shading launches of a step on several streams

    // scene
    bool have_scene = false, have_camera = false;
    // Staging buffers of the query entry points (first hits, light pdfs, material probes, camera rays): kept from call
    // to call and only ever grown, so that a caller that polls does not pay a hipMalloc / hipFree pair per call;
    // a request above STAGE_KEEP bytes is served by an allocation of its own and given back at once.
    static constexpr size_t STAGE_KEEP = 256u << 20;
    struct Stage {
        void *p = nullptr;
        size_t bytes = 0;
    } stage[3];
    std::vector<std::pair<void *, size_t>> scene_allocs;
    DevScene scene{};
    uint32_t n_queues = 0;
    int queue_type[MAX_QUEUES];  // material type served by queue slot q

    // camera + film
    DevCamera cam{};
    float *accum = nullptr;
    uint8_t *rgb8 = nullptr;
    uint32_t n_samples = 0;
    uint64_t film_pixels = 0;  // pixels `accum` and `rgb8` are allocated for (0: none)

    // Path state.  A render call is cut into batches of strata, and up to n_pipes batches are
    // in flight at once, each on its own stream with its own path state: the late bounces of
    // one batch (few rays, long dependent chains) then overlap the early bounces of another.
    // pipes[0] runs on `stream` and also serves the first-hit and per-sample entry points.
    struct Pipe {
        hipStream_t stream = nullptr;
        // The shading launches of one step (one per material type) are independent of each other: those
        // after the first go to these streams, between a fork and a join on `stream`, so that a
        // launch with few hits does not have the chip to itself.
        static constexpr int N_SIDE = 2;
        hipStream_t side[N_SIDE] = {nullptr, nullptr};
        hipEvent_t fork = nullptr, join[N_SIDE] = {nullptr, nullptr};
        PathState ps{};
        std::vector<std::pair<void *, size_t>> allocs;
        uint64_t capacity = 0;
        hipEvent_t accumulated[2] = {nullptr, nullptr};  // recorded after a batch's k_accumulate, alternately
        uint64_t batches = 0;                            // batches enqueued on this pipe so far
        uint32_t *host_chunks[2] = {nullptr, nullptr};   // pinned: rays per bounce of the batches before
        uint32_t host_chunks_depth[2] = {0, 0};          // bounces recorded in each (0: nothing yet)
        uint32_t host_paths[2] = {0, 0};                 // paths of the batch each belongs to
        // profiling: raygen, per-bounce extend/shade, accumulate; two sets used alternately so that
        // reading one back never waits for the batch just enqueued
        std::vector<hipEvent_t> events[2];
        uint32_t events_pending[2] = {0, 0};  // bounces whose events have not been read back yet
    };
    // Rolling schedule: every launch of a pipe carries bounce 0 of the batch started with it, bounce 1
    // of the batch before, and so on, so that no launch is thin however few paths reach the late
    // bounces.  Batch b of the pipe lives in region b % n_regions of the slot arrays and is
    // complete max_depth launches ("ticks") after it started.
    static constexpr int ROLL_RING = 8;  // ticks the host may run ahead of a pipe
    struct Due {
        uint64_t tick;  // the tick whose shading finishes the batch
        uint32_t region, n_strata;
    };
    struct Roll {
        PathState ps{};
        std::vector<std::pair<void *, size_t>> allocs;
        uint32_t n_regions = 0, cap = 0;
        uint64_t n_live = 0;  // rays in flight the arrays have room for
        uint64_t tick = 0;
        // Admission: a batch starts only if its paths and the rays already in flight fit n_live.  The shading of
        // tick t counts the rays it queues for tick t + 1 (tick_rays[t % ROLL_RING], copied to host_tick_rays when
        // the tick is through); ticks [0, counted) are known to be through, known_rays = what the last of them
        // queued, and ring_new[] holds the paths started by the ticks after that.
        uint32_t *tick_rays = nullptr;
        uint32_t *host_tick_rays = nullptr;  // pinned
        uint32_t ring_new[ROLL_RING] = {};
        uint64_t counted = 0;
        uint32_t known_rays = 0;
        uint64_t finishes = 0;  // finishing scans on this path state so far (they take its two counters in turn)
        // How fast this scene's paths end: drain_rays[j] = rays still queued after tick last_start + j, where
        // last_start is the last tick that started a batch (so nothing but survivors is counted).
        uint64_t last_start = 0;
        bool any_start = false;
        std::vector<uint32_t> drain_rays;
        std::deque<Due> due;
        hipEvent_t done[ROLL_RING] = {}, accumulated[ROLL_RING] = {};
        // before raygen, before extend, after extend, after shading; around the camera-ray kernel on its side stream (4, 5);
        // after the ray-drawing kernel, before it waits for the other (6)
        hipEvent_t prof[ROLL_RING][7] = {};
        bool prof_pending[ROLL_RING] = {}, prof_packet[ROLL_RING] = {}, prof_walk[ROLL_RING] = {};
    };
    static constexpr int MAX_PIPES = 8;
    Pipe pipes[MAX_PIPES];
    Roll rolls[MAX_PIPES];
    uint64_t roll_seq = 0;        // ticks over all pipes so far (picks the pipe: strictly round-robin)
    int64_t roll_k = -1;             // strata per tick decided for the current scene, camera and options (-1: not yet)
    uint32_t roll_seen = 0;          // ... of which so many are used: the largest number one pt_render call has asked for so far
    uint64_t untraced_paths = 0;  // camera paths of renders with max_depth 0, which start and end on the host
    uint64_t roll_paths = 32000000;  // paths started per tick; 0 = one batch at a time per pipe
    int n_pipes = 4;       // batch-at-a-time schedule
    int n_roll_pipes = 1;  // rolling schedule: with large launches a second pipe gains 0-3 % (or loses) and doubles the path state;
    // launches below ROLL_SMALL_LAUNCH paths do not fill the chip through their last rays, and a second pipe's launches
    // run beside them: measured +12 % at 4.2 M paths per launch (Cornell box 256 x 256 x 64), +17 % at 4.7 M and +18 % at
    // 9.4 M (knob scene), +8 % and +15 % at 8.4 M (1 M-triangle box, interior), -10 % at 18.9 M (knob scene).  Chosen
    // when the launch size is known, unless PT_OPT_PIPES has said how many.
    bool roll_pipes_auto = true;
    static constexpr uint64_t ROLL_SMALL_LAUNCH = 12000000;
    uint64_t batch_seq = 0;                 // batches enqueued so far (picks the pipe)
    hipEvent_t last_accumulated = nullptr;  // of the most recent batch: the next k_accumulate waits for it
    // Rays per bounce of the most recent finished batch ([0] = its paths): launches of later batches are
    // sized from it (any size is correct, the kernels loop over their queues; this only avoids
    // dispatching thousands of workgroups that find nothing to do in the thin late bounces).
    std::vector<uint32_t> bounce_profile;
    DevStats *stats = nullptr;

    // profiling
    hipEvent_t epoch = nullptr;                      // time zero of the current render call
    std::vector<std::pair<float, float>> extend_iv;  // (start, end) of its closest-hit launches, ms since epoch
    double extend_busy_ms = 0;                       // time during which at least one closest-hit launch was running
    std::vector<double> bounce_extend_ms, bounce_shade_ms;  // per bounce, summed over batches (profiling)
    uint64_t profiled_batches = 0;
    double host_wait_ms = 0;  // part of render_ms spent waiting so as not to run too far ahead of the device
    double extend_ms = 0, shade_ms = 0, other_ms = 0, render_ms = 0, walk_ms = 0, packet_ms = 0;
    uint64_t walk_launches = 0, packet_launches = 0;
    uint64_t device_bytes = 0;
};

static thread_local std::string g_create_error;

static int fail(pt_context *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                      \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return fail(ctx, PT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <class T>
static int dev_alloc(pt_context *ctx, std::vector<std::pair<void *, size_t>> &pool, T **out, size_t count) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(ctx, PT_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    pool.push_back({p, bytes});
    ctx->device_bytes += bytes;
    *out = (T *)p;
    return PT_OK;
}
template <class T>
static int dev_upload(pt_context *ctx, std::vector<std::pair<void *, size_t>> &pool, T **out, const T *src, size_t count) {
    int rc = dev_alloc(ctx, pool, out, count);
    if (rc) return rc;
    if (count) HIP_TRY(ctx, hipMemcpy(*out, src, count * sizeof(T), hipMemcpyHostToDevice));
    return PT_OK;
}
static void free_pool(pt_context *ctx, std::vector<std::pair<void *, size_t>> &pool) {
    for (auto &p : pool) {
        (void)hipFree(p.first);
        ctx->device_bytes -= p.second;
    }
    pool.clear();
}

// Device memory of at least `bytes` for staging slot i of a query entry point; *transient: the caller frees it.
static int stage_get(pt_context *ctx, int i, size_t bytes, void **out, bool *transient) {
    pt_context::Stage &s = ctx->stage[i];
    *transient = false;
    if (bytes <= s.bytes) {
        *out = s.p;
        return PT_OK;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(ctx, PT_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    if (bytes > pt_context::STAGE_KEEP) {
        *transient = true;
    } else {
        if (s.p) (void)hipFree(s.p);
        ctx->device_bytes += bytes - s.bytes;
        s.p = p;
        s.bytes = bytes;
    }
    *out = p;
    return PT_OK;
}
// Up to three staging buffers at once; put() gives back the ones that were too large to keep.
struct Staging {
    pt_context *ctx;
    void *p[3] = {nullptr, nullptr, nullptr};
    bool transient[3] = {false, false, false};
    explicit Staging(pt_context *c) : ctx(c) {}
    template <class T>
    int get(int i, size_t count, T **out) {
        void *q = nullptr;
        int rc = stage_get(ctx, i, std::max<size_t>(count, 1) * sizeof(T), &q, &transient[i]);
        if (rc) return rc;
        p[i] = q;
        *out = (T *)q;
        return PT_OK;
    }
    ~Staging() {
        for (int i = 0; i < 3; i++)
            if (transient[i] && p[i]) (void)hipFree(p[i]);
    }
};

// workgroups of a traversal kernel that fit a CU's 160 KB of LDS with their lane stacks
constexpr int MAX_WALK_BLOCKS = 163840 / (LDS_LEVELS * BLOCK * 8 + 1024);
static int grid_for(const pt_context *ctx, uint64_t n) {
    uint64_t blocks = (n + BLOCK - 1) / BLOCK;
    uint64_t cap = (uint64_t)ctx->n_cu * 8;
    return (int)std::max<uint64_t>(1, std::min(blocks, cap));
}
// Kernels whose element count lives in device memory are launched at a fixed size.
static int grid_fixed(const pt_context *ctx, uint64_t upper_bound, int blocks_per_cu = 0) {
    uint64_t blocks = (upper_bound + BLOCK - 1) / BLOCK;
    uint64_t cap = (uint64_t)ctx->n_cu * (blocks_per_cu ? blocks_per_cu : ctx->blocks_per_cu);  // 5 workgroups of 32 KB LDS fit one CU
    blocks = std::max<uint64_t>(1, std::min(blocks, cap));
    return (int)((blocks + 7) / 8 * 8);  // the same number of workgroups on every XCD
}

static hipError_t scratch_lookup(int device);

// Streams, events and pinned words of pipe i, made when the pipe is first used: the defaults use one pipe, and a
// context that made all eight at creation spent most of its set-up time on them.
static hipError_t pipe_init(pt_context *ctx, int i) {
    pt_context::Pipe &pp = ctx->pipes[i];
    pt_context::Roll &r = ctx->rolls[i];
    if (pp.fork) return hipSuccess;
    hipError_t e = hipSuccess;
    if (!pp.stream) e = hipStreamCreate(&pp.stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pp.fork, hipEventDisableTiming);
    for (int k = 0; k < pt_context::Pipe::N_SIDE && e == hipSuccess; k++) {
        e = hipStreamCreateWithFlags(&pp.side[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pp.join[k], hipEventDisableTiming);
    }
    for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipEventCreateWithFlags(&pp.accumulated[k], hipEventDisableTiming);
    for (int k = 0; k < 2 && e == hipSuccess; k++) e = hipHostMalloc((void **)&pp.host_chunks[k], BOUNCE_PROFILE_LEN * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&r.host_tick_rays, pt_context::ROLL_RING * sizeof(uint32_t), hipHostMallocDefault);
    for (int k = 0; k < pt_context::ROLL_RING && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&r.done[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.accumulated[k], hipEventDisableTiming);
        for (int j = 0; j < 7 && e == hipSuccess; j++) e = hipEventCreate(&r.prof[k][j]);
    }
    return e;
}

static void free_roll(pt_context *ctx, pt_context::Roll &r) {
    free_pool(ctx, r.allocs);
    r.n_regions = r.cap = 0;
    r.n_live = 0;
    r.tick_rays = nullptr;
    r.counted = 0;
    r.known_rays = 0;
    r.tick = 0;
    r.due.clear();
    for (bool &b : r.prof_pending) b = false;
}
static void destroy_pipes(pt_context *ctx) {
    for (auto &r : ctx->rolls) {
        free_roll(ctx, r);
        if (r.host_tick_rays) (void)hipHostFree(r.host_tick_rays);
        r.host_tick_rays = nullptr;
        for (int i = 0; i < pt_context::ROLL_RING; i++) {
            if (r.done[i]) (void)hipEventDestroy(r.done[i]);
            if (r.accumulated[i]) (void)hipEventDestroy(r.accumulated[i]);
            r.done[i] = r.accumulated[i] = nullptr;
            for (auto &ev : r.prof[i]) {
                if (ev) (void)hipEventDestroy(ev);
                ev = nullptr;
            }
        }
    }
    for (auto &pp : ctx->pipes) {
        if (pp.stream) (void)hipStreamSynchronize(pp.stream);
        free_pool(ctx, pp.allocs);
        pp.capacity = 0;
        for (auto &set : pp.events) {
            for (hipEvent_t ev : set) (void)hipEventDestroy(ev);
            set.clear();
        }
        for (auto &ev : pp.accumulated) {
            if (ev) (void)hipEventDestroy(ev);
            ev = nullptr;
        }
        for (auto &hc : pp.host_chunks) {
            if (hc) (void)hipHostFree(hc);
            hc = nullptr;
        }
        for (int k = 0; k < pt_context::Pipe::N_SIDE; k++) {
            if (pp.side[k]) (void)hipStreamDestroy(pp.side[k]);
            if (pp.join[k]) (void)hipEventDestroy(pp.join[k]);
            pp.side[k] = nullptr;
            pp.join[k] = nullptr;
        }
        if (pp.fork) (void)hipEventDestroy(pp.fork);
        pp.fork = nullptr;
        if (pp.stream) (void)hipStreamDestroy(pp.stream);
        pp.stream = nullptr;
    }
    ctx->stream = nullptr;
    if (ctx->epoch) (void)hipEventDestroy(ctx->epoch);
    ctx->epoch = nullptr;
}
static int roll_flush(pt_context *ctx);
static uint32_t roll_learn(pt_context *ctx);
static int ensure_roll(pt_context *ctx, pt_context::Pipe &pp, pt_context::Roll &r, uint32_t n_regions, uint32_t cap, uint64_t n_live);
// Waits for everything enqueued (finish_batches: after enqueueing what the batches under way still need).
static void sync_pipes(pt_context *ctx, bool finish_batches = true) {
    if (finish_batches) (void)roll_flush(ctx);
    for (auto &pp : ctx->pipes)
        if (pp.stream) (void)hipStreamSynchronize(pp.stream);
}
// What earlier batches looked like says nothing about the next ones (new scene, camera or sampling).
static void forget_bounce_profile(pt_context *ctx) {
    sync_pipes(ctx);
    ctx->bounce_profile.clear();
    for (auto &pp : ctx->pipes) pp.host_chunks_depth[0] = pp.host_chunks_depth[1] = 0;
}

// The entry points below have C linkage through their declarations in ptcore.h.
static void harvest_profile(pt_context *ctx, pt_context::Pipe &pp);
static void harvest_roll(pt_context *ctx, pt_context::Roll &r, int ring);

// Waits for everything enqueued on the context and folds in pending event timings.
static int sync_all(pt_context *ctx) {
    if (int rc = roll_flush(ctx)) return rc;
    for (int i = 0; i < pt_context::MAX_PIPES; i++) {
        auto &pp = ctx->pipes[i];
        if (!pp.stream) continue;
        HIP_TRY(ctx, hipStreamSynchronize(pp.stream));
        harvest_profile(ctx, pp);
        for (int ring = 0; ring < pt_context::ROLL_RING; ring++) harvest_roll(ctx, ctx->rolls[i], ring);
    }
    if (!ctx->extend_iv.empty()) {  // union of the closest-hit launches since the last call
        std::sort(ctx->extend_iv.begin(), ctx->extend_iv.end());
        float lo = ctx->extend_iv[0].first, hi = ctx->extend_iv[0].second;
        for (const auto &iv : ctx->extend_iv) {
            if (iv.first > hi) {
                ctx->extend_busy_ms += hi - lo;
                lo = iv.first;
                hi = iv.second;
            } else {
                hi = std::max(hi, iv.second);
            }
        }
        ctx->extend_busy_ms += hi - lo;
        ctx->extend_iv.clear();
    }
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->batch_ticks == 0u) {
        // what the drain has shown about this scene's paths; the slot arrays change size now, while nothing is under way
        const uint32_t learnt = roll_learn(ctx);
        if (learnt != 0u && (ctx->auto_ticks == 0u || learnt > ctx->auto_ticks || learnt + 4u <= ctx->auto_ticks)) {
            ctx->auto_ticks = learnt;
            const uint32_t regions = ctx->roll_regions();
            for (int p = 0; p < ctx->n_roll_pipes; p++) {
                pt_context::Roll &r = ctx->rolls[p];
                if (r.cap != 0u && r.n_regions != regions)
                    if (int rc = ensure_roll(ctx, ctx->pipes[p], r, regions, r.cap, r.n_live)) return rc;
            }
        }
    }
    {
        unsigned long long dropped = 0;
        HIP_TRY(ctx, hipMemcpy(&dropped, &ctx->stats->queue_overflow, sizeof dropped, hipMemcpyDeviceToHost));
        if (dropped) {
            HIP_TRY(ctx, hipMemset(&ctx->stats->queue_overflow, 0, sizeof dropped));
            return fail(ctx, PT_ERR_STATE, "a work queue ran out of room and %llu paths were dropped: the image is incomplete", dropped);
        }
    }
#ifdef PT_BOUNDS_CHECK
    unsigned long long bad[8];
    HIP_TRY(ctx, hipMemcpy(bad, ctx->stats->bad_refs, sizeof bad, hipMemcpyDeviceToHost));
    if (bad[0] || bad[1])
        return fail(ctx, PT_ERR_STATE,
                    "device checks: %llu bad references, %llu lanes gave up; first: check %llu phase %llu cur %#llx node_base %llu li %llu lend %llu "
                    "inst %#llx prim_base %llu sp %llu top %llu slot %llu child %llu",
                    bad[0], bad[1], bad[2] >> 32, bad[2] & 0xFFFFFFFFull, bad[3] & 0xFFFFFFFFull, bad[3] >> 32, bad[4] >> 32, bad[4] & 0xFFFFFFFFull,
                    bad[5] & 0xFFFFFFFFull, bad[5] >> 32, bad[6] & 0xFFFFFFFFull, bad[6] >> 32, bad[7] & 0xFFFFFFFFull, bad[7] >> 32);
#endif
    return PT_OK;
}
// Time zero for the event intervals of the profile.
static int restart_epoch(pt_context *ctx) {
    int rc = sync_all(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->epoch, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->epoch));
    return PT_OK;
}

int pt_sync(pt_context *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return sync_all(ctx);
}

int pt_create(int device, pt_context **out) {
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, PT_ERR_HIP, "no HIP device available: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, PT_ERR_INVALID, "device %d out of range (have %d)", device, n);
    pt_context *ctx = new (std::nothrow) pt_context();
    if (!ctx) return fail(nullptr, PT_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    // PT_DEBUG_CREATE: where the time of this call goes, to stderr
    const bool timing = getenv("PT_DEBUG_CREATE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pt_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PT_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    }
    lap("hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) == hipSuccess) ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    lap("hipGetDeviceProperties");
    if ((e = scratch_lookup(device)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PT_ERR_HIP, "hipGetSymbolAddress of the static scratch array on device %d: %s", device, hipGetErrorString(e));
    }
    lap("static array address");
    if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, PT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    lap("hipStreamCreate");
    ctx->pipes[0].stream = ctx->stream;
    e = hipEventCreate(&ctx->epoch);
    if (e == hipSuccess) e = pipe_init(ctx, 0);  // the other pipes get their streams and events when first used
    lap("streams, events, pinned words");
    void *p = nullptr;
    if (e != hipSuccess || (e = hipMalloc(&p, sizeof(DevStats))) != hipSuccess) {
        destroy_pipes(ctx);
        delete ctx;
        return fail(nullptr, PT_ERR_NOMEM, "stream/event/stat allocation: %s", hipGetErrorString(e));
    }
    ctx->stats = (DevStats *)p;
    (void)hipMemset(ctx->stats, 0, sizeof(DevStats));
    lap("counters");
    *out = ctx;
    return PT_OK;
}

void pt_destroy(pt_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    sync_pipes(ctx, false);
    free_pool(ctx, ctx->scene_allocs);
    if (ctx->accum) (void)hipFree(ctx->accum);
    if (ctx->rgb8) (void)hipFree(ctx->rgb8);
    if (ctx->stats) (void)hipFree(ctx->stats);
    for (auto &st : ctx->stage)
        if (st.p) (void)hipFree(st.p);
    destroy_pipes(ctx);
    delete ctx;
}

const char *pt_last_error(const pt_context *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_set_option(pt_context *ctx, int key, double value) {
    if (!ctx) return PT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    sync_pipes(ctx);  // batches under way finish under the settings they started with
    ctx->roll_k = -1;
    switch (key) {
        case PT_OPT_BSDF_SAMPLE_PROBABILITY:
            if (!(value >= 0.0 && value <= 1.0)) return fail(ctx, PT_ERR_INVALID, "bsdf sample probability must be in [0,1]");
            ctx->bsdf_prob = (float)value;
            ctx->auto_ticks = 0;  // paths end at another rate
            forget_bounce_profile(ctx);
            return PT_OK;
        case PT_OPT_MAX_PATHS:
            if (!(value >= 1.0 && value <= 4.0e9)) return fail(ctx, PT_ERR_INVALID, "max paths out of range");
            ctx->max_paths = (uint64_t)value;
            return PT_OK;
        case PT_OPT_PROFILE: {
            int rc = restart_epoch(ctx);
            if (rc) return rc;
            ctx->profile = value != 0.0;
            return PT_OK;
        }
        case PT_OPT_COUNTERS: ctx->counters = value != 0.0; return PT_OK;
        case PT_OPT_TRAVERSAL:
            if (value != 0.0 && value != 1.0 && value != 2.0) return fail(ctx, PT_ERR_INVALID, "traversal must be 0, 1 or 2");
            ctx->force_ref_order = value == 1.0;
            ctx->trace_packet = value == 2.0;
            return PT_OK;
        case PT_OPT_PIPES:
            if (!(value >= 1.0 && value <= (double)pt_context::MAX_PIPES)) return fail(ctx, PT_ERR_INVALID, "pipes must be in [1,%d]", pt_context::MAX_PIPES);
            sync_pipes(ctx);
            ctx->n_pipes = ctx->n_roll_pipes = (int)value;
            ctx->roll_pipes_auto = false;
            return PT_OK;
        case PT_OPT_REFILL_BELOW:
            if (!(value >= 1.0 && value <= 64.0)) return fail(ctx, PT_ERR_INVALID, "refill threshold must be in [1,64]");
            ctx->refill_below = (int)value;
            return PT_OK;
        case PT_OPT_BLOCKS_PER_CU:
            if (!(value >= 1.0 && value <= (double)MAX_WALK_BLOCKS))
                return fail(ctx, PT_ERR_INVALID, "workgroups per CU must be in [1,%d] (LDS holds that many traversal stacks)", MAX_WALK_BLOCKS);
            sync_pipes(ctx);
            ctx->blocks_per_cu = (int)value;
            for (auto &pp : ctx->pipes) {  // the stack overflow area is sized by the launch width
                free_pool(ctx, pp.allocs);
                pp.capacity = 0;
            }
            for (auto &r : ctx->rolls) free_roll(ctx, r);
            return PT_OK;
        case PT_OPT_INTEGRATOR:
            if (value != 0.0 && value != 1.0) return fail(ctx, PT_ERR_INVALID, "integrator must be 0 (the reference's) or 1 (next-event estimation)");
            ctx->nee = value != 0.0;
            ctx->auto_ticks = 0;
            for (auto &pp : ctx->pipes) {  // the shadow-ray queue exists only with next-event estimation
                free_pool(ctx, pp.allocs);
                pp.capacity = 0;
            }
            for (auto &r : ctx->rolls) free_roll(ctx, r);
            forget_bounce_profile(ctx);
            return PT_OK;
        case PT_OPT_DEVICE_BVH:
            if (!(value >= 0.0 && value <= 4.0e9)) return fail(ctx, PT_ERR_INVALID, "triangle count out of range");
            ctx->device_bvh_min = (uint32_t)value;
            return PT_OK;
        case PT_OPT_ROLLING:
            if (!(value >= 0.0 && value <= 1.0e9)) return fail(ctx, PT_ERR_INVALID, "paths per launch out of range");
            ctx->roll_paths = (uint64_t)value;
            return PT_OK;
        case PT_OPT_TAIL_BELOW:  // kept for callers that set it: the launch it switched on is gone
            if (!(value >= 0.0 && value <= 1.0e9)) return fail(ctx, PT_ERR_INVALID, "tail threshold out of range");
            return PT_OK;
        case PT_OPT_PACKET:
            if (value != 0.0 && value != 1.0) return fail(ctx, PT_ERR_INVALID, "packet option must be 0 or 1");
            sync_pipes(ctx);
            ctx->packet = value != 0.0;
            return PT_OK;
        case PT_OPT_BATCH_TICKS:
            if (!(value >= 0.0 && value <= 1.0e6) || value != (double)(uint32_t)value) return fail(ctx, PT_ERR_INVALID, "batch ticks must be a count");
            ctx->batch_ticks = (uint32_t)value;
            return PT_OK;
        case PT_OPT_SHADE_CLAIM:
            if (!(value >= -1.0 && value <= 64.0) || value != (double)(int)value) return fail(ctx, PT_ERR_INVALID, "shade claim must be -1, 0 or a number of chunks up to 64");
            ctx->shade_claim = (int)value;
            return PT_OK;
        case PT_OPT_LIVE_FACTOR:
            if (!(value >= 2.0 && value <= 1.0e6)) return fail(ctx, PT_ERR_INVALID, "live factor must be at least 2");
            ctx->live_factor = value;
            return PT_OK;
        case PT_OPT_LEAF_BATCH:
            if (!(value >= 1.0 && value <= 64.0)) return fail(ctx, PT_ERR_INVALID, "leaf batch must be in [1,64]");
            ctx->leaf_batch = (int)value;
            return PT_OK;
    }
    return fail(ctx, PT_ERR_INVALID, "unknown option %d", key);
}

// ---------------------------------------------------------------- scene validation + upload
static bool buffer_ok(const pt_buffer_desc &b, uint32_t n_prims, uint32_t n_nodes) {
    return (uint64_t)b.prim_first + b.prim_count <= n_prims && (uint64_t)b.node_first + b.node_count <= n_nodes;
}

// Checks one flat node array: children lie after their parent and inside the array, leaf
// ranges lie inside the buffer, and no root-to-leaf path is longer than 64 nodes (the
// traversal stack is sized from that; the reference's builder refuses deeper trees).
static bool nodes_ok(const pt_node *nodes, uint32_t count, uint32_t prim_count, std::string &why) {
    std::vector<std::pair<uint32_t, uint32_t>> stack;  // (node, depth)
    stack.push_back({0u, 1u});
    uint64_t visited = 0;
    while (!stack.empty()) {
        auto [i, depth] = stack.back();
        stack.pop_back();
        if (i >= count) { why = "child index outside the node array"; return false; }
        if (depth > 64) { why = "BVH deeper than 64 levels"; return false; }
        if (++visited > count) { why = "node array is not a tree"; return false; }
        const pt_node &nd = nodes[i];
        if (nd.b & PT_NODE_LEAF) {
            const uint32_t size = nd.b & 0x7FFFFFFFu;
            if (size > 0xFFFFu) { why = "leaf size above 65535"; return false; }
            if ((uint64_t)nd.a + size > prim_count) { why = "leaf range outside its primitive buffer"; return false; }
        } else {
            if (nd.b > 2u) { why = "split axis above 2"; return false; }
            if (nd.a <= i + 1 || nd.a >= count) { why = "right child must follow the left subtree"; return false; }
            stack.push_back({nd.a, depth + 1});
            stack.push_back({i + 1, depth + 1});
        }
    }
    return true;
}

// Converts one reference-order node array into 64-byte two-box nodes, appended to `wide`.
// The first nodes of the result are the top of the tree in breadth-first order (they are
// what every ray touches); below that each subtree is laid out depth-first so that a
// descent walks forward through memory.  Returns false when a leaf does not fit the packed
// reference (no primitive, more than 127 primitives, or an offset of 2^24 or more).
struct WideBuild {
    uint32_t wide_first, root_ref, depth;
    float rmin[3], rmax[3];
};
// The two boxes of a 64-byte node (dev_traverse.h, wide_box_l / wide_box_r read them back).
static void wide_boxes(float4 *w, const float *lmin, const float *lmax, const float *rmin, const float *rmax) {
    w[0] = make_float4(lmin[0], lmin[1], lmax[0], lmax[1]);
    w[1] = make_float4(lmin[2], lmax[2], rmin[2], rmax[2]);
    w[2] = make_float4(rmin[0], rmin[1], rmax[0], rmax[1]);
}
static bool build_wide(const pt_node *nodes, uint32_t count, std::vector<float4> &wide, WideBuild &out) {
    out.wide_first = (uint32_t)(wide.size() / 4);
    std::memcpy(out.rmin, nodes[0].bmin, 12);
    std::memcpy(out.rmax, nodes[0].bmax, 12);
    out.depth = 1;
    auto leaf_ref = [](const pt_node &n, uint32_t &ref) {
        const uint32_t size = n.b & 0x7FFFFFFFu;
        if (size == 0u || size > 127u || n.a >= (1u << 24)) return false;
        ref = WIDE_LEAF | (size << 24) | n.a;
        return true;
    };
    if (nodes[0].b & PT_NODE_LEAF) return leaf_ref(nodes[0], out.root_ref);
    // 1. order of interior nodes
    constexpr size_t TOP = 255;
    std::vector<uint32_t> order, queue{0};
    std::vector<uint32_t> depth_of(count, 0);
    depth_of[0] = 1;
    size_t qh = 0;
    while (qh < queue.size() && order.size() < TOP) {
        const uint32_t i = queue[qh++];
        order.push_back(i);
        const uint32_t kids[2] = {i + 1, nodes[i].a};
        for (uint32_t c : kids) {
            depth_of[c] = depth_of[i] + 1;
            if (!(nodes[c].b & PT_NODE_LEAF)) queue.push_back(c);
        }
    }
    std::vector<uint32_t> stack;
    for (; qh < queue.size(); qh++) {  // remaining frontier: depth-first below each
        stack.push_back(queue[qh]);
        while (!stack.empty()) {
            const uint32_t i = stack.back();
            stack.pop_back();
            order.push_back(i);
            const uint32_t l = i + 1, r = nodes[i].a;
            depth_of[l] = depth_of[r] = depth_of[i] + 1;
            if (!(nodes[r].b & PT_NODE_LEAF)) stack.push_back(r);
            if (!(nodes[l].b & PT_NODE_LEAF)) stack.push_back(l);
        }
    }
    std::vector<uint32_t> wide_index(count, 0xFFFFFFFFu);
    for (size_t k = 0; k < order.size(); k++) wide_index[order[k]] = (uint32_t)k;
    for (uint32_t i = 0; i < count; i++) out.depth = std::max(out.depth, depth_of[i]);
    // 2. fill
    const size_t base = wide.size();
    wide.resize(base + 4 * order.size());
    for (size_t k = 0; k < order.size(); k++) {
        const uint32_t i = order[k];
        const pt_node &L = nodes[i + 1], &R = nodes[nodes[i].a];
        uint32_t lref, rref;
        if (L.b & PT_NODE_LEAF) { if (!leaf_ref(L, lref)) return false; } else lref = wide_index[i + 1];
        if (R.b & PT_NODE_LEAF) { if (!leaf_ref(R, rref)) return false; } else rref = wide_index[nodes[i].a];
        float4 *w = &wide[base + 4 * k];
        wide_boxes(w, L.bmin, L.bmax, R.bmin, R.bmax);
        float fl, fr, fa;
        const uint32_t axis = nodes[i].b;
        std::memcpy(&fl, &lref, 4);
        std::memcpy(&fr, &rref, 4);
        std::memcpy(&fa, &axis, 4);
        w[3] = make_float4(fl, fr, fa, 0.0f);
    }
    out.root_ref = 0;  // the root is the first node of the breadth-first part
    return true;
}

int pt_upload_scene(pt_context *ctx, const pt_scene_desc *s) {
    if (!ctx || !s) return fail(ctx, PT_ERR_INVALID, "pt_upload_scene: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (s->n_sets < 2 || !s->sets) return fail(ctx, PT_ERR_INVALID, "scene needs sets[0] (objects) and sets[1] (lights)");
    if ((s->n_tris && (!s->tri_data || !s->tri_mat)) || (s->n_spheres && (!s->spheres || !s->sphere_mat)) ||
        (s->n_quads && (!s->quads || !s->quad_mat)) || (s->n_instances && !s->instances) || (s->n_nodes && !s->nodes) ||
        (s->n_materials && !s->materials) || (s->n_textures && !s->textures) || (s->n_images && !s->images) ||
        (s->n_texel_bytes && !s->texels))
        return fail(ctx, PT_ERR_INVALID, "scene array pointer is NULL while its count is not zero");
    std::string why;
    for (uint32_t k = 0; k < s->n_sets; k++) {
        const pt_primset_desc &ps = s->sets[k];
        if (!buffer_ok(ps.tris, s->n_tris, s->n_nodes) || !buffer_ok(ps.spheres, s->n_spheres, s->n_nodes) ||
            !buffer_ok(ps.quads, s->n_quads, s->n_nodes) || !buffer_ok(ps.instances, s->n_instances, s->n_nodes))
            return fail(ctx, PT_ERR_INVALID, "set %u: range outside the scene arrays", k);
        const pt_buffer_desc *bufs[4] = {&ps.tris, &ps.spheres, &ps.quads, &ps.instances};
        for (int t = 0; t < 4; t++)
            if (bufs[t]->node_count && !nodes_ok(s->nodes + bufs[t]->node_first, bufs[t]->node_count, bufs[t]->prim_count, why))
                return fail(ctx, PT_ERR_INVALID, "set %u buffer %d: %s", k, t, why.c_str());
        if (k >= 1 && ps.instances.prim_count && k != 1)
            return fail(ctx, PT_ERR_UNSUPPORTED, "set %u: instances inside an instance: list them in the world's set with their levels as chain records (libpthost does)", k);
    }
    for (uint32_t i = 0; i < s->n_instances; i++) {
        const pt_instance_desc &in = s->instances[i];
        if (in.child_set < 2 || in.child_set >= s->n_sets) return fail(ctx, PT_ERR_INVALID, "instance %u: child_set out of range", i);
        if ((uint64_t)in.chain_first + in.chain_count > s->n_instances) return fail(ctx, PT_ERR_INVALID, "instance %u: chain records out of range", i);
        for (uint32_t k = 0; k < in.chain_count; k++)
            if (s->instances[in.chain_first + k].chain_count != 0) return fail(ctx, PT_ERR_INVALID, "instance %u: a chain record has a chain of its own", i);
        const float *r3t = in.transform + 12, *r3i = in.inverse + 12;
        if (r3t[0] != 0.0f || r3t[1] != 0.0f || r3t[2] != 0.0f || r3t[3] != 1.0f || r3i[0] != 0.0f || r3i[1] != 0.0f || r3i[2] != 0.0f || r3i[3] != 1.0f)
            return fail(ctx, PT_ERR_UNSUPPORTED, "instance %u: matrix is not affine", i);
    }
    auto mat_ok = [&](const int32_t *m, uint32_t n) {
        for (uint32_t i = 0; i < n; i++)
            if (m[i] < -1 || m[i] >= (int32_t)s->n_materials) return false;
        return true;
    };
    if (!mat_ok(s->tri_mat, s->n_tris) || !mat_ok(s->sphere_mat, s->n_spheres) || !mat_ok(s->quad_mat, s->n_quads))
        return fail(ctx, PT_ERR_INVALID, "material index out of range");
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const pts_texture &t = s->textures[i];
        if (t.type > PTS_TEX_NOISE) return fail(ctx, PT_ERR_INVALID, "texture %u: unknown type", i);
        if (t.type == PTS_TEX_CHECKERED) {
            // children must come earlier in the table: rules out cycles
            if (t.odd < 0 || t.even < 0 || t.odd >= (int32_t)i || t.even >= (int32_t)i)
                return fail(ctx, PT_ERR_INVALID, "texture %u: checker children must be defined before it", i);
        }
        if (t.type == PTS_TEX_IMAGE && t.image >= 0) {
            if (t.image >= (int32_t)s->n_images) return fail(ctx, PT_ERR_INVALID, "texture %u: image index out of range", i);
            const pts_image &im = s->images[t.image];
            if (im.offset + 3ull * im.width * im.height > s->n_texel_bytes) return fail(ctx, PT_ERR_INVALID, "image %d: texels out of range", t.image);
        }
    }
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const pt_material_desc &m = s->materials[i];
        if (m.type >= PTS_MAT_COUNT) return fail(ctx, PT_ERR_INVALID, "material %u: unknown type", i);
        const bool needs_tex = !(m.type == PTS_MAT_METAL || m.type == PTS_MAT_DIELECTRIC || m.type == PTS_MAT_DISNEY_CLEARCOAT);
        if (needs_tex && (m.texture < 0 || m.texture >= (int32_t)s->n_textures)) return fail(ctx, PT_ERR_INVALID, "material %u: texture index out of range", i);
    }
    if (s->n_media > PTS_MAX_MEDIA || (s->n_media && !s->media)) return fail(ctx, PT_ERR_INVALID, "at most %d participating media, in a non-NULL array", (int)PTS_MAX_MEDIA);
    for (uint32_t i = 0; i < s->n_media; i++) {
        const pts_medium &m = s->media[i];
        if (!(m.radius > 0.0f) || !(m.density > 0.0f) || m.texture < 0 || m.texture >= (int32_t)s->n_textures)
            return fail(ctx, PT_ERR_INVALID, "medium %u: radius and density must be positive and the texture must exist", i);
    }
    if (s->n_moving > PTS_MAX_MOVING || s->n_moving > s->n_spheres || (s->n_moving && !s->moving))
        return fail(ctx, PT_ERR_INVALID, "at most %d moving spheres, the last of spheres[], with their motion in a non-NULL array", (int)PTS_MAX_MOVING);
    for (uint32_t k = 0; k < s->n_sets; k++)
        if ((uint64_t)s->sets[k].spheres.prim_first + s->sets[k].spheres.prim_count > s->n_spheres - s->n_moving)
            return fail(ctx, PT_ERR_INVALID, "set %u: moving spheres belong to no set", k);
    // a shade queue (and a shade kernel per bounce) only for material types some primitive uses
    bool type_present[MAX_QUEUES] = {false};
    type_present[PTD_MAT_MEDIUM] = s->n_media != 0;
    auto mark = [&](const int32_t *m, uint32_t n) {
        for (uint32_t i = 0; i < n; i++)
            if (m[i] >= 0) type_present[s->materials[m[i]].type] = true;
    };
    mark(s->tri_mat, s->n_tris);
    mark(s->sphere_mat, s->n_spheres);
    mark(s->quad_mat, s->n_quads);

    // ---- build device images
    forget_bounce_profile(ctx);
    free_pool(ctx, ctx->scene_allocs);
    ctx->have_scene = false;
    DevScene d{};
    std::vector<float4> hot((size_t)s->n_tris * 3), cold((size_t)s->n_tris * 6);
    for (uint32_t i = 0; i < s->n_tris; i++) {
        const float *t = s->tri_data + 24 * (size_t)i;  // t[row*3 + vertex]
        const float v0x = t[0], v0y = t[3], v0z = t[6];
        const float e1x = t[1] - v0x, e1y = t[4] - v0y, e1z = t[7] - v0z;  // position(1) - position(0)
        const float e2x = t[2] - v0x, e2y = t[5] - v0y, e2z = t[8] - v0z;  // position(2) - position(0)
        hot[3 * (size_t)i] = make_float4(v0x, v0y, v0z, e1x);
        hot[3 * (size_t)i + 1] = make_float4(e1y, e1z, e2x, e2y);
        hot[3 * (size_t)i + 2] = make_float4(e2z, 0.0f, 0.0f, 0.0f);
        std::memcpy(&cold[6 * (size_t)i], t, 24 * sizeof(float));
    }
    std::vector<float4> sph(s->n_spheres), quads((size_t)s->n_quads * 4), insts((size_t)s->n_instances * 7), nodes((size_t)s->n_nodes * 2);
    std::vector<float> qarea(s->n_quads);
    for (uint32_t i = 0; i < s->n_spheres; i++) sph[i] = make_float4(s->spheres[4 * i], s->spheres[4 * i + 1], s->spheres[4 * i + 2], s->spheres[4 * i + 3]);
    for (uint32_t i = 0; i < s->n_quads; i++) {
        std::memcpy(&quads[4 * (size_t)i], s->quads + 17 * (size_t)i, 16 * sizeof(float));
        qarea[i] = s->quads[17 * (size_t)i + 16];
    }
    for (uint32_t i = 0; i < s->n_instances; i++) {
        float *o = (float *)&insts[7 * (size_t)i];
        std::memcpy(o, s->instances[i].inverse, 12 * sizeof(float));
        std::memcpy(o + 12, s->instances[i].transform, 12 * sizeof(float));
        uint32_t cs = s->instances[i].child_set;
        std::memcpy(o + 24, &cs, 4);
        std::memcpy(o + 25, &s->instances[i].chain_first, 4);
        std::memcpy(o + 26, &s->instances[i].chain_count, 4);
        o[27] = 0.0f;
    }
    static_assert(sizeof(pt_node) == 32, "pt_node must be 32 bytes");
    if (s->n_nodes) std::memcpy(nodes.data(), s->nodes, (size_t)s->n_nodes * 32);
    std::vector<DevSet> sets(s->n_sets);
    std::vector<float4> wide;
    bool wide_ok = true;
    // depth of the BVH of each buffer: [set][triangles, spheres, quads, instances]
    std::vector<std::array<uint32_t, 4>> depth_of(s->n_sets, std::array<uint32_t, 4>{0, 0, 0, 0});
    struct DeviceBuild {
        uint32_t set, n, prim_first, wide_first;
    };
    std::vector<DeviceBuild> builds;
    auto cvt = [&](const pt_buffer_desc &b, uint32_t k, int type) {
        DevBuffer d{};
        d.prim_first = b.prim_first;
        d.prim_count = b.prim_count;
        d.node_first = b.node_first;
        d.node_count = b.node_count;
        if (b.node_count) {
            WideBuild wb{};
            if (!build_wide(s->nodes + b.node_first, b.node_count, wide, wb)) wide_ok = false;
            d.wide_first = wb.wide_first;
            d.root_ref = wb.root_ref;
            std::memcpy(d.rmin, wb.rmin, 12);
            std::memcpy(d.rmax, wb.rmax, 12);
            d.tree = 1;
            depth_of[k][type] = wb.depth;
        } else if (type == 0 && k != 1 && ctx->device_bvh_min && b.prim_count >= ctx->device_bvh_min) {  // lights are never traversed
            d.wide_first = (uint32_t)(wide.size() / 4);
            wide.resize(wide.size() + 4 * (size_t)(b.prim_count - 1), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            d.tree = 2;  // root and box follow from the build below
            builds.push_back({k, b.prim_count, b.prim_first, d.wide_first});
        }
        return d;
    };
    for (uint32_t k = 0; k < s->n_sets; k++)
        sets[k] = DevSet{cvt(s->sets[k].tris, k, 0), cvt(s->sets[k].spheres, k, 1), cvt(s->sets[k].quads, k, 2), cvt(s->sets[k].instances, k, 3)};
    for (const DeviceBuild &b : builds)
        if (b.n > (1u << 24)) return fail(ctx, PT_ERR_UNSUPPORTED, "set %u: more than 2^24 triangles in one buffer", b.set);
    bool boxes_finite = true;
    for (uint32_t i = 0; i < s->n_nodes && boxes_finite; i++)
        for (int k = 0; k < 3; k++) boxes_finite = boxes_finite && std::isfinite(s->nodes[i].bmin[k]) && std::isfinite(s->nodes[i].bmax[k]);
    std::vector<DevMaterial> mats(s->n_materials);
    static_assert(sizeof(DevMaterial) == sizeof(pt_material_desc), "material layout");
    if (s->n_materials) std::memcpy(mats.data(), s->materials, (size_t)s->n_materials * sizeof(DevMaterial));
    ctx->n_queues = 0;
    int slot_of_type[MAX_QUEUES];
    for (int t = 0; t < MAX_QUEUES; t++) {
        slot_of_type[t] = -1;
        if (type_present[t]) {
            slot_of_type[t] = (int)ctx->n_queues;
            ctx->queue_type[ctx->n_queues++] = t;
        }
    }
    std::vector<uint8_t> mat_queue(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; i++) mat_queue[i] = (uint8_t)(slot_of_type[s->materials[i].type] < 0 ? 0xFF : slot_of_type[s->materials[i].type]);

    int rc;
    float4 *p4;
#define UP4(field, vec)                                                            \
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &p4, vec.data(), vec.size()))) return rc; \
    d.field = p4;
    UP4(nodes, nodes) UP4(wide, wide) UP4(tri_hot, hot) UP4(tri_cold, cold) UP4(spheres, sph) UP4(quads, quads) UP4(insts, insts)
#undef UP4
    int32_t *pi;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pi, s->tri_mat, s->n_tris))) return rc;
    d.tri_mat = pi;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pi, s->sphere_mat, s->n_spheres))) return rc;
    d.sphere_mat = pi;
    {
        std::vector<uint32_t> identity(s->n_tris);
        for (uint32_t i = 0; i < s->n_tris; i++) identity[i] = i;
        uint32_t *po;
        if ((rc = dev_upload(ctx, ctx->scene_allocs, &po, identity.data(), identity.size()))) return rc;
        d.tri_order = po;
        for (const DeviceBuild &b : builds) {
            TriBvhResult res{};
            std::string err;
            if (build_tri_bvh_device(ctx->stream, const_cast<float4 *>(d.tri_hot) + 3 * (size_t)b.prim_first, const_cast<float4 *>(d.tri_cold) + 6 * (size_t)b.prim_first,
                                     const_cast<int32_t *>(d.tri_mat) + b.prim_first, b.n, const_cast<float4 *>(d.wide) + 4 * (size_t)b.wide_first, po + b.prim_first, &res,
                                     err))
                return fail(ctx, PT_ERR_HIP, "device BVH build of set %u: %s", b.set, err.c_str());
            DevBuffer &t = sets[b.set].tris;
            t.root_ref = res.root_ref;
            std::memcpy(t.rmin, res.rmin, 12);
            std::memcpy(t.rmax, res.rmax, 12);
            depth_of[b.set][0] = res.height;
            for (int k = 0; k < 3; k++) boxes_finite = boxes_finite && std::isfinite(res.rmin[k]) && std::isfinite(res.rmax[k]);
        }
    }
    // stack entries a ray can hold at once: one per level of the BVH it is in, plus, inside
    // an instance, those of the instance BVH above it
    {
        auto set_depth = [&](uint32_t k) { return std::max(depth_of[k][0], std::max(depth_of[k][1], depth_of[k][2])); };
        uint32_t child_depth = 0;
        for (uint32_t k = 2; k < s->n_sets; k++) child_depth = std::max(child_depth, set_depth(k));
        // + the newest entry kept in registers, + the two records an instance descent leaves on the stack
        ctx->stack_need = std::max(set_depth(0), depth_of[0][3] + child_depth) + 4u;
        ctx->wide_ok = wide_ok;
        const DevSet &w = sets[0];
        ctx->packet_ok = true;  // every set a ray can get into has reference-order nodes (or no tree at all)
        for (const DevSet &ds : sets) ctx->packet_ok = ctx->packet_ok && ds.tris.tree != 2u && ds.spheres.tree != 2u && ds.quads.tree != 2u && ds.insts.tree != 2u;
        // ... and it enters an instance in one step: the levels of an instance that was nested are the per-lane kernel's
        for (uint32_t i = 0; i < s->n_instances; i++) ctx->packet_ok = ctx->packet_ok && s->instances[i].chain_count == 0;
        (void)w;
    }
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pi, s->quad_mat, s->n_quads))) return rc;
    d.quad_mat = pi;
    float *pf;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pf, qarea.data(), qarea.size()))) return rc;
    d.quad_area = pf;
    DevSet *pset;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pset, sets.data(), sets.size()))) return rc;
    d.sets = pset;
    DevMaterial *pm;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pm, mats.data(), mats.size()))) return rc;
    d.mats = pm;
    uint8_t *pu8;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pu8, mat_queue.data(), mat_queue.size()))) return rc;
    d.mat_queue = pu8;
    pts_texture *pt;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pt, s->textures, s->n_textures))) return rc;
    d.textures = pt;
    pts_image *pim;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pim, s->images, s->n_images))) return rc;
    d.images = pim;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pu8, s->texels, (size_t)s->n_texel_bytes))) return rc;
    d.texels = pu8;
    pts_medium *pmed;
    if ((rc = dev_upload(ctx, ctx->scene_allocs, &pmed, s->media, s->n_media))) return rc;
    d.media = pmed;
    d.n_media = s->n_media;
    d.medium_queue = s->n_media ? (uint32_t)slot_of_type[PTD_MAT_MEDIUM] : 0xFFu;
    {
        std::vector<float4> delta(s->n_moving);
        for (uint32_t i = 0; i < s->n_moving; i++) delta[i] = make_float4(s->moving[3 * i], s->moving[3 * i + 1], s->moving[3 * i + 2], 0.0f);
        float4 *pd;
        if ((rc = dev_upload(ctx, ctx->scene_allocs, &pd, delta.data(), delta.size()))) return rc;
        d.moving_delta = pd;
        d.n_moving = s->n_moving;
        d.n_static_spheres = s->n_spheres - s->n_moving;
    }
    {
        // A box around every light, for light_pdf to skip the lights a ray cannot hit.  The exact tests
        // (Triangle::hit, Sphere::hit, Quad::hit in f32) accept rays whose exact line passes a little
        // outside the primitive.  For triangles and quads that is a few ulp of (|origin| + |light|).  A
        // sphere's discriminant h^2 - |d|^2 (|oc|^2 - r^2) carries an absolute error of up to about
        // 13 * 2^-24 |d|^2 |oc|^2, so a ray may miss a small, distant sphere by sqrt(8e-7) |oc| and still
        // count as a hit; with |oc| <= sqrt(3) (|origin| + |light|) in largest-coordinate terms that is
        // 1.6e-3 (|origin| + |light|).  The boxes are widened by more than that.
        const float rel_flat = 2.0e-5f, rel_sphere = 4.0e-3f;
        const DevSet &L = sets[1];
        std::vector<float4> boxes;
        auto add_box = [&](const float *lo, const float *hi, float rel, bool finite) {
            if (!finite) {  // never skipped
                boxes.push_back(make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f));
                boxes.push_back(make_float4(INFINITY, INFINITY, INFINITY, 0.0f));
                return;
            }
            float m = 0.0f;
            for (int k = 0; k < 3; k++) m = std::max(m, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
            const float pad = rel * m + 1.0e-30f;
            boxes.push_back(make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, 0.0f));
            boxes.push_back(make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.0f));
        };
        auto all_finite = [](const float *p, int n) {
            bool ok = true;
            for (int k = 0; k < n; k++) ok = ok && std::isfinite(p[k]);
            return ok;
        };
        for (uint32_t i = 0; i < L.tris.prim_count; i++) {
            const float *t = s->tri_data + 24 * (size_t)(L.tris.prim_first + i);
            float lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                lo[k] = std::min(t[3 * k], std::min(t[3 * k + 1], t[3 * k + 2]));
                hi[k] = std::max(t[3 * k], std::max(t[3 * k + 1], t[3 * k + 2]));
            }
            add_box(lo, hi, rel_flat, all_finite(t, 9));
        }
        for (uint32_t i = 0; i < L.spheres.prim_count; i++) {
            const float *c = s->spheres + 4 * (size_t)(L.spheres.prim_first + i);
            const float r = std::fabs(c[3]);
            const float lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
            add_box(lo, hi, rel_sphere, all_finite(c, 4));
        }
        for (uint32_t i = 0; i < L.quads.prim_count; i++) {
            const float *q = s->quads + 17 * (size_t)(L.quads.prim_first + i);  // origin, u, v, ...
            float lo[3], hi[3];
            for (int k = 0; k < 3; k++) {
                const float a = q[k], b = q[k] + q[3 + k], c = q[k] + q[6 + k], e = (q[k] + q[3 + k]) + q[6 + k];
                lo[k] = std::min(std::min(a, b), std::min(c, e));
                hi[k] = std::max(std::max(a, b), std::max(c, e));
            }
            add_box(lo, hi, rel_flat, all_finite(q, 9));
        }
        // and around every 32 consecutive lights of a kind
        std::vector<float4> groups;
        size_t at = 0;
        for (uint32_t n : {L.tris.prim_count, L.spheres.prim_count, L.quads.prim_count}) {
            for (uint32_t g = 0; g < n; g += 32u) {
                float4 lo = boxes[2 * (at + g)], hi = boxes[2 * (at + g) + 1];
                for (uint32_t j = g + 1; j < std::min(n, g + 32u); j++) {
                    const float4 a = boxes[2 * (at + j)], b = boxes[2 * (at + j) + 1];
                    lo = make_float4(std::min(lo.x, a.x), std::min(lo.y, a.y), std::min(lo.z, a.z), 0.0f);
                    hi = make_float4(std::max(hi.x, b.x), std::max(hi.y, b.y), std::max(hi.z, b.z), 0.0f);
                }
                groups.push_back(lo);
                groups.push_back(hi);
            }
            at += n;
        }
        float4 *pb;
        if ((rc = dev_upload(ctx, ctx->scene_allocs, &pb, boxes.data(), boxes.size()))) return rc;
        d.light_boxes = pb;
        if ((rc = dev_upload(ctx, ctx->scene_allocs, &pb, groups.data(), groups.size()))) return rc;
        d.light_group_boxes = pb;
        d.light_pad_flat = rel_flat;
        d.light_pad_sphere = rel_sphere;
    }
    d.world = sets[0];
    d.lights = sets[1];
    d.n_materials = s->n_materials;
    d.n_light_total = sets[1].tris.prim_count + sets[1].spheres.prim_count + sets[1].quads.prim_count + sets[1].insts.prim_count;
    d.boxes_finite = boxes_finite ? 1u : 0u;
    d.n_wide = (uint32_t)(wide.size() / 4);
    d.n_tris = s->n_tris;
    d.n_spheres = s->n_spheres;
    d.n_quads = s->n_quads;
    d.n_insts = s->n_instances;
    d.n_sets = s->n_sets;
    ctx->scene = d;
    ctx->have_scene = true;
    // shade queues depend on which material types exist: drop path state so it is rebuilt
    for (auto &pp : ctx->pipes) {
        free_pool(ctx, pp.allocs);
        pp.capacity = 0;
    }
    for (auto &r : ctx->rolls) free_roll(ctx, r);
    ctx->roll_k = -1;
    ctx->auto_ticks = 0;
    return PT_OK;
}

int pt_set_camera(pt_context *ctx, const pt_camera_desc *c) {
    if (!ctx || !c) return fail(ctx, PT_ERR_INVALID, "pt_set_camera: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (c->image_width == 0 || c->image_height == 0 || (uint64_t)c->image_width * c->image_height > (1ull << 28))
        return fail(ctx, PT_ERR_INVALID, "image size out of range");
    if (c->sqrt_spp == 0) return fail(ctx, PT_ERR_INVALID, "sqrt_spp must be at least 1");
    DevCamera d{};
    d.width = c->image_width;
    d.height = c->image_height;
    auto v = [](const float *p) { return f3{p[0], p[1], p[2]}; };
    d.center = v(c->center);
    d.pixel00 = v(c->pixel00);
    d.pixel_du = v(c->pixel_du);
    d.pixel_dv = v(c->pixel_dv);
    d.disk_u = v(c->defocus_disk_u);
    d.disk_v = v(c->defocus_disk_v);
    d.defocus_angle = c->defocus_angle;
    d.sqrt_spp = c->sqrt_spp;
    d.inv_sqrt_spp = c->inv_sqrt_spp;
    d.max_depth = c->max_depth;
    d.background = v(c->background);
    forget_bounce_profile(ctx);
    const uint64_t npix = (uint64_t)d.width * d.height;
    // until buffers of the new size exist and are cleared the context has no camera
    ctx->have_camera = false;
    if (ctx->film_pixels != npix) {
        if (ctx->accum) (void)hipFree(ctx->accum);
        if (ctx->rgb8) (void)hipFree(ctx->rgb8);
        ctx->accum = nullptr;
        ctx->rgb8 = nullptr;
        ctx->device_bytes -= ctx->film_pixels * 15;
        ctx->film_pixels = 0;
        hipError_t e = hipMalloc((void **)&ctx->accum, npix * 3 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&ctx->rgb8, npix * 3);
        if (e != hipSuccess) {
            if (ctx->accum) (void)hipFree(ctx->accum);
            ctx->accum = nullptr;
            return fail(ctx, PT_ERR_NOMEM, "hipMalloc of the accumulator and image: %s", hipGetErrorString(e));
        }
        ctx->film_pixels = npix;
        ctx->device_bytes += npix * 15;
    }
    HIP_TRY(ctx, hipMemset(ctx->accum, 0, npix * 3 * sizeof(float)));
    ctx->n_samples = 0;
    ctx->cam = d;
    ctx->have_camera = true;
    ctx->roll_k = -1;
    ctx->auto_ticks = 0;
    return PT_OK;
}

// ---------------------------------------------------------------- path state
// Queue capacities in chunks for path state of n_slots paths.  A closed chunk holds more than QUEUE_CHUNK - 64
// entries, and every producing wave of every launch that fills a queue may leave one open chunk behind: a
// material queue is filled by the closest-hit launches of a step (two side by side when the camera rays have
// their own kernel), a ray queue (and the shadow-ray queue) by one shading launch per material type in use.
static void queue_capacity(const pt_context *ctx, uint64_t n_slots, uint64_t &max_chunks, uint64_t &max_ray_chunks) {
    const uint64_t max_waves = (uint64_t)ctx->n_cu * std::max({ctx->blocks_per_cu, ctx->shade_blocks_per_cu, 8}) * (BLOCK / 64);
    // the material queues of a step are filled by the ray-drawing kernel and, beside it, by the camera-ray kernel
    // (8 workgroups per CU): each wave of either may leave one open chunk per queue
    const uint64_t packet_waves = (uint64_t)ctx->n_cu * 8u * (BLOCK / 64);
    max_chunks = n_slots / (QUEUE_CHUNK - 63u) + max_waves + packet_waves + 8u;
    max_ray_chunks = n_slots / (QUEUE_CHUNK - 63u) + (uint64_t)std::max<uint32_t>(1u, ctx->n_queues) * max_waves + 8u;
}

// Path state for `n_slots` paths of which at most `n_live` have a ray queued for one step and at most `n_fresh`
// start with one step (dev_scene.h, PathState): radiance by slot, travelling state by ray-queue position, hits
// and material queues by work-list position.
static int alloc_path_state(pt_context *ctx, std::vector<std::pair<void *, size_t>> &pool, hipStream_t st, PathState &ps, uint64_t n_slots, uint64_t n_live,
                            uint64_t n_fresh, uint32_t n_regions) {
    int rc;
    uint64_t unused, max_ray_chunks, max_chunks, max_shadow_chunks;
    queue_capacity(ctx, n_live, unused, max_ray_chunks);
    queue_capacity(ctx, n_live + n_fresh, max_chunks, max_shadow_chunks);
    const uint64_t positions = max_ray_chunks * QUEUE_CHUNK;
    if (positions + n_fresh >= (1ull << 32) || max_shadow_chunks * QUEUE_CHUNK >= (1ull << 32)) return fail(ctx, PT_ERR_INVALID, "too many rays in flight for 32-bit positions");
    ps.max_chunks = (uint32_t)max_chunks;
    ps.max_ray_chunks = (uint32_t)max_ray_chunks;
    ps.queue_overflow = &ctx->stats->queue_overflow;
    ps.n_queues = ctx->n_queues;
    if ((rc = dev_alloc(ctx, pool, &ps.contrib, n_slots))) return rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = dev_alloc(ctx, pool, &ps.ray_o[k], positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.ray_d[k], positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.beta[k], positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.ray_queue[k], positions))) return rc;
    }
    if ((rc = dev_alloc(ctx, pool, &ps.hit, positions + n_fresh))) return rc;
    if ((rc = dev_alloc(ctx, pool, &ps.hit2, positions + n_fresh))) return rc;
    if ((rc = dev_alloc(ctx, pool, &ps.fresh_draws, n_fresh + 1u))) return rc;
    for (uint32_t q = 0; q < ctx->n_queues; q++)
        if ((rc = dev_alloc(ctx, pool, &ps.shade_queue[q], max_chunks * QUEUE_CHUNK))) return rc;
    {
        // PT_FINISH_LIST_CAP (tests only): a short list, so that the scan for what did not fit is exercised
        const char *env = getenv("PT_FINISH_LIST_CAP");
        const uint64_t want = env ? (uint64_t)std::max(1L, atol(env)) : (1ull << 20);
        ps.finish_cap = (uint32_t)std::min<uint64_t>(want, std::max<uint64_t>(n_slots, 1));
        if ((rc = dev_alloc(ctx, pool, &ps.finish_list, (size_t)ps.finish_cap))) return rc;
    }
    if ((rc = dev_alloc(ctx, pool, &ps.ctl, (size_t)CTL_WORDS))) return rc;
    if ((rc = dev_alloc(ctx, pool, &ps.regions, (size_t)n_regions))) return rc;
    // radiance slots start at zero and are left at zero by whoever reads them (k_accumulate, pt_render_samples)
    HIP_TRY(ctx, hipMemsetAsync(ps.contrib, 0, n_slots * sizeof(float4), st));
    HIP_TRY(ctx, hipMemsetAsync(ps.ctl, 0, CTL_WORDS * sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(ps.regions, 0, n_regions * sizeof(RegionParams), st));
    ps.moving = ctx->scene.n_moving != 0 ? 1u : 0u;
    ps.nee = ctx->nee ? 1u : 0u;
    if (ctx->nee) {  // at most one shadow ray per vertex, filled by every shading launch of a step
        const uint64_t shadow_positions = max_shadow_chunks * QUEUE_CHUNK;
        if ((rc = dev_alloc(ctx, pool, &ps.shadow_queue, shadow_positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.shadow_o, shadow_positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.shadow_d, shadow_positions))) return rc;
        if ((rc = dev_alloc(ctx, pool, &ps.shadow_c, shadow_positions))) return rc;
    }
    // stack entries beyond the LDS part: one slab of (lanes in the widest launch) per level
    ps.overflow_stride = (uint32_t)grid_fixed(ctx, ~0ull >> 1) * BLOCK;
    const uint32_t extra = ctx->stack_need > (uint32_t)LDS_LEVELS ? ctx->stack_need - LDS_LEVELS : 0u;
    if ((rc = dev_alloc(ctx, pool, &ps.overflow, (size_t)extra * ps.overflow_stride))) return rc;
    ps.refill_below = ctx->refill_below;
    ps.leaf_batch = ctx->leaf_batch;
    ps.shade_claim = 0u;
    ps.shade_claim_from = 0u;
    ps.n_slots = (uint32_t)n_slots;
    return PT_OK;
}
static int ensure_paths(pt_context *ctx, pt_context::Pipe &pp, uint64_t n_slots) {
    if (n_slots <= pp.capacity) return PT_OK;
    (void)hipStreamSynchronize(pp.stream);
    free_pool(ctx, pp.allocs);
    pp.capacity = 0;
    PathState ps{};
    // one batch at a time: every path may be alive at every bounce, and all of them start with the first step
    int rc = alloc_path_state(ctx, pp.allocs, pp.stream, ps, n_slots, n_slots, n_slots, 1u);
    if (rc) return rc;
    if ((rc = dev_alloc(ctx, pp.allocs, &ps.bounce_rays, (size_t)BOUNCE_PROFILE_LEN))) return rc;
    ps.next_rays = nullptr;
    pp.ps = ps;
    pp.capacity = n_slots;
    return PT_OK;
}

static hipEvent_t get_event(pt_context::Pipe &pp, size_t i) {
    auto &set = pp.events[pp.batches & 1u];
    while (set.size() <= i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        set.push_back(ev);
    }
    return set[i];
}

// One closest-hit step.  Counting AABB / primitive tests implies the reference-order loop.
// fp: the paths that start with this step (on their camera rays; nothing is stored for them).
// side, fork, join: a second stream of the pipe with two events (or null): where k_extend_packet runs.
// queued: about how many rays the ray queue holds (the last count the host has seen).
// prof (or null): three events: around the camera-ray kernel on its stream, and after the ray-drawing kernel; *packet_used: whether
// the camera-ray kernel was launched.
static void launch_extend(pt_context *ctx, hipStream_t st, int grid, const PathState &ps, uint32_t bounce, FreshPaths fp, hipStream_t side = nullptr,
                          hipEvent_t fork = nullptr, hipEvent_t join = nullptr, uint64_t queued = 0, hipEvent_t *prof = nullptr, bool *packet_used = nullptr) {
    if (packet_used) *packet_used = false;
    dim3 g(grid);
    const dim3 b(BLOCK);
    const f3 bg = ctx->cam.background;
    const bool insts = ctx->scene.world.insts.prim_count != 0;
#define EXT(C, K) hipLaunchKernelGGL((k_extend<C, K>), g, b, 0, st, ctx->scene, ctx->cam, ps, bounce, bg, fp, ctx->stats)
#define EXI(C) hipLaunchKernelGGL((k_extend_inst<C>), g, b, 0, st, ctx->scene, ctx->cam, ps, bounce, bg, fp, ctx->stats)
    if (ctx->counters) {
        if (insts) { EXT(true, false); EXI(true); } else EXT(true, true);
    } else if (ctx->use_wide()) {
        FreshPaths walk_fp = fp;
        if (ctx->use_packet() && fp.n_new != 0u && side != nullptr) {
            // the new paths a wave at a time, beside the rays queued by earlier steps
            (void)hipEventRecord(fork, st);
            (void)hipStreamWaitEvent(side, fork, 0);
            // The two kernels share a CU only if the walking one leaves registers free: five of its waves per SIMD
            // take 480 of the 512 VGPRs, and a wave of this one needs 64.  Where most of the step's rays are new
            // paths the walking kernel runs with four; where the queued rays are the larger part it keeps five
            // (1 M-triangle box, interior: 0 to -3 % with four, knob: +9 %).
            const bool mostly_new = (uint64_t)fp.n_new >= queued;
            // the camera-ray kernel hands its work out piece by piece, so its grid is what a CU can hold at once
            const unsigned per_cu = 8u;
            const dim3 gp((unsigned)std::min<uint64_t>(((uint64_t)fp.n_new + BLOCK - 1) / BLOCK, (uint64_t)ctx->n_cu * per_cu));
#define EXP(I, M) hipLaunchKernelGGL((k_extend_packet<I, M>), gp, b, 0, side, ctx->scene, ctx->cam, ps, bounce, bg, fp, ctx->stats)
            if (prof) (void)hipEventRecord(prof[0], side);
            if (ctx->scene.n_moving != 0) {
                if (insts) EXP(true, true); else EXP(false, true);
            } else {
                if (insts) EXP(true, false); else EXP(false, false);
            }
#undef EXP
            if (prof) (void)hipEventRecord(prof[1], side);
            if (packet_used) *packet_used = true;
            (void)hipEventRecord(join, side);
            walk_fp.n_new = 0u;
            const int beside = mostly_new ? 4 : 5;
            if (beside < ctx->blocks_per_cu) g = dim3(std::min<unsigned>(g.x, (unsigned)grid_fixed(ctx, ~0ull >> 1, beside)));
        }
#define EXW(I, M, F, T) hipLaunchKernelGGL((k_extend_walk<I, M, F, T>), g, b, 0, st, ctx->scene, ctx->cam, ps, bounce, bg, walk_fp, ctx->stats)
#define EXWF(I, M, T) do { if (walk_fp.n_new != 0u) EXW(I, M, true, T); else EXW(I, M, false, T); } while (0)
        const bool tail = !insts && walk_tail_applies(ctx->scene.world);  // its few spheres and quads need no search
        if (ctx->scene.n_moving != 0) {
            if (insts) EXWF(true, true, false); else if (tail) EXWF(false, true, true); else EXWF(false, true, false);
        } else {
            if (insts) EXWF(true, false, false); else if (tail) EXWF(false, false, true); else EXWF(false, false, false);
        }
#undef EXWF
#undef EXW
        if (prof) (void)hipEventRecord(prof[2], st);
        if (walk_fp.n_new != fp.n_new) (void)hipStreamWaitEvent(st, join, 0);
    } else {
        if (insts) { EXT(false, false); EXI(false); } else EXT(false, true);
    }
#undef EXT
#undef EXI
}

static void launch_shadow(pt_context *ctx, hipStream_t st, int grid, const PathState &ps, uint32_t bounce) {
    if (ctx->scene.world.insts.prim_count != 0) hipLaunchKernelGGL((k_shadow_walk<true>), dim3(grid), dim3(BLOCK), 0, st, ctx->scene, ps, bounce, ctx->stats);
    else hipLaunchKernelGGL((k_shadow_walk<false>), dim3(grid), dim3(BLOCK), 0, st, ctx->scene, ps, bounce, ctx->stats);
}

struct ShadeLaunch {
    pt_context *ctx;
    hipStream_t st;
    const PathState *ps;
    int grid;
    const BatchParams *bp;
    uint32_t bounce, q;
    FreshPaths fp;
};
template <int MT>
static void launch_shade(const ShadeLaunch &l) {
#define SHADE(N) hipLaunchKernelGGL((k_shade<MT, N>), dim3(l.grid), dim3(BLOCK), 0, l.st, l.ctx->scene, l.ctx->cam, *l.ps, *l.bp, l.bounce, l.q, l.fp, l.ctx->stats)
    if (l.ps->nee && l.ctx->scene.n_media != 0) SHADE(3);
    else if (l.ps->nee) SHADE(1);
    else if (l.ctx->scene.n_media != 0) SHADE(2);
    else SHADE(0);
#undef SHADE
}
static void launch_shade_type(int type, const ShadeLaunch &l);
// The shading launches of one step, one per material type in the scene, spread over the pipe's streams.
static int launch_shades(pt_context *ctx, pt_context::Pipe &pp, ShadeLaunch l) {
    constexpr int N_SIDE = pt_context::Pipe::N_SIDE;
    const hipStream_t st = pp.stream;
    const bool spread = ctx->shade_spread && ctx->n_queues > 1;
    if (spread) {
        HIP_TRY(ctx, hipEventRecord(pp.fork, st));
        for (int k = 0; k < N_SIDE && k + 1 < (int)ctx->n_queues; k++) HIP_TRY(ctx, hipStreamWaitEvent(pp.side[k], pp.fork, 0));
    }
    for (uint32_t q = 0; q < ctx->n_queues; q++) {
        const int lane = spread ? (int)(q % (N_SIDE + 1)) : 0;
        l.st = lane == 0 ? st : pp.side[lane - 1];
        l.q = q;
        launch_shade_type(ctx->queue_type[q], l);
    }
    if (spread)
        for (int k = 0; k < N_SIDE && k + 1 < (int)ctx->n_queues; k++) {
            HIP_TRY(ctx, hipEventRecord(pp.join[k], pp.side[k]));
            HIP_TRY(ctx, hipStreamWaitEvent(st, pp.join[k], 0));
        }
    return PT_OK;
}
static void launch_shade_type(int type, const ShadeLaunch &l) {
    switch (type) {
        case PTS_MAT_LAMBERTIAN: launch_shade<PTS_MAT_LAMBERTIAN>(l); break;
        case PTS_MAT_METAL: launch_shade<PTS_MAT_METAL>(l); break;
        case PTS_MAT_DIELECTRIC: launch_shade<PTS_MAT_DIELECTRIC>(l); break;
        case PTS_MAT_DIFFUSE_LIGHT: launch_shade<PTS_MAT_DIFFUSE_LIGHT>(l); break;
        case PTS_MAT_DISNEY: launch_shade<PTS_MAT_DISNEY>(l); break;
        case PTS_MAT_DISNEY_DIFFUSE: launch_shade<PTS_MAT_DISNEY_DIFFUSE>(l); break;
        case PTS_MAT_DISNEY_METAL: launch_shade<PTS_MAT_DISNEY_METAL>(l); break;
        case PTS_MAT_DISNEY_GLASS: launch_shade<PTS_MAT_DISNEY_GLASS>(l); break;
        case PTS_MAT_DISNEY_CLEARCOAT: launch_shade<PTS_MAT_DISNEY_CLEARCOAT>(l); break;
        case PTS_MAT_DISNEY_SHEEN: launch_shade<PTS_MAT_DISNEY_SHEEN>(l); break;
        case PTD_MAT_MEDIUM: launch_shade<PTD_MAT_MEDIUM>(l); break;
    }
}

// Under light sampling a path can outlive max_depth (a vertex that samples a light resets the depth left to 2,
// src/camera.rs:332): what is left of a batch when it is due is finished by k_finish.
static bool paths_may_outlive_depth(const pt_context *ctx) { return !ctx->nee && ctx->bsdf_prob < 1.0f; }
// k: 0 / 1 in turn for the calls on one path state (which of its two counters the scan uses).
static void launch_finish(pt_context *ctx, hipStream_t st, const PathState &ps, const BatchParams &bp, uint32_t bounce, uint32_t region, uint32_t k) {
    const dim3 b(BLOCK);
    // the scan with every wave the chip holds; the listed paths (a handful, as a rule) with one workgroup per CU, and
    // so the scan for what the list could not hold, which as a rule finds nothing to do and returns (a larger grid
    // of that 256-register kernel costs 30 us a step just to start and end)
    hipLaunchKernelGGL(k_finish_scan, dim3((unsigned)ctx->n_cu * 8u), b, 0, st, ps, bounce, region, k);
    const dim3 gl((unsigned)std::min<uint64_t>((uint64_t)ctx->n_cu, ((uint64_t)ps.finish_cap + BLOCK - 1) / BLOCK));
#define FIN(M, L, G) hipLaunchKernelGGL((k_finish<M, L>), G, b, 0, st, ctx->scene, ctx->cam, ps, bp, bounce, region, ctx->cam.background, k, ctx->stats)
    const dim3 g((unsigned)ctx->n_cu);
    if (ctx->scene.n_media != 0) { FIN(true, true, gl); FIN(true, false, g); }
    else { FIN(false, true, gl); FIN(false, false, g); }
#undef FIN
}

static uint64_t mix64_host(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Reads back the event timings a pipe still holds (profiling only); waits for that batch.
static void harvest_set(pt_context *ctx, pt_context::Pipe &pp, int which) {
    if (!pp.events_pending[which]) return;
    const uint32_t depth = pp.events_pending[which];
    pp.events_pending[which] = 0;
    const auto &evs = pp.events[which];
    const size_t last = 3 + 2 * (size_t)depth;
    if (evs.size() <= last || hipEventSynchronize(evs[last]) != hipSuccess) return;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, evs[0], evs[1]);
    ctx->other_ms += ms;
    for (uint32_t b = 0; b < depth; b++) {
        (void)hipEventElapsedTime(&ms, evs[1 + 2 * b], evs[2 + 2 * b]);
        ctx->extend_ms += ms;
        if (ctx->bounce_extend_ms.size() <= b) {
            ctx->bounce_extend_ms.resize(b + 1, 0.0);
            ctx->bounce_shade_ms.resize(b + 1, 0.0);
        }
        ctx->bounce_extend_ms[b] += ms;
        float since = 0;
        if (hipEventElapsedTime(&since, ctx->epoch, evs[1 + 2 * b]) == hipSuccess) ctx->extend_iv.push_back({since, since + ms});
        (void)hipEventElapsedTime(&ms, evs[2 + 2 * b], evs[3 + 2 * b]);
        ctx->shade_ms += ms;
        ctx->bounce_shade_ms[b] += ms;
    }
    ctx->profiled_batches++;
    (void)hipEventElapsedTime(&ms, evs[last - 1], evs[last]);
    ctx->other_ms += ms;
}
static void harvest_profile(pt_context *ctx, pt_context::Pipe &pp) {
    harvest_set(ctx, pp, 0);
    harvest_set(ctx, pp, 1);
}

// Enqueues one batch (strata [first, first + k)) on a pipe: every bounce, no host wait.
// Radiance is left in the pipe's ps.contrib.
// *bounces_out: bounces that have their own pair of profiling events (the tail launch counts as one).
static int run_batch(pt_context *ctx, pt_context::Pipe &pp, uint32_t first, uint32_t k, uint64_t seed, uint32_t *bounces_out = nullptr) {
    const uint32_t npix = ctx->cam.width * ctx->cam.height;
    const uint64_t n_slots = (uint64_t)npix * k;
    harvest_set(ctx, pp, (int)(pp.batches & 1u));  // the set this batch is about to record into
    int rc = ensure_paths(ctx, pp, n_slots);
    if (rc) return rc;
    BatchParams bp{mix64_host(seed), first, k, npix, (uint32_t)n_slots, ctx->nee ? 1.0f : ctx->bsdf_prob, ctx->cam.background};
    hipStream_t st = pp.stream;
    HIP_TRY(ctx, hipMemsetAsync(pp.ps.ctl, 0, CTL_WORDS * sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(pp.ps.bounce_rays, 0, BOUNCE_PROFILE_LEN * sizeof(uint32_t), st));
    pp.ps.n_slots = (uint32_t)n_slots;
    pp.ps.region_cap = (uint32_t)n_slots;  // one region: the batch
    pp.ps.refill_below = ctx->refill_below;
    pp.ps.leaf_batch = ctx->leaf_batch;
    pp.ps.shade_claim = ctx->shade_claim > 0 ? (uint32_t)ctx->shade_claim : 0u;  // chosen step by step only by the rolling schedule
    pp.ps.shade_claim_from = 0u;
    const PathState ps = pp.ps;
    RegionParams rp{};
    rp.seed_mixed = bp.seed_mixed;
    rp.stratum_first = first;
    rp.n_strata = k;
    hipLaunchKernelGGL(k_tick_begin, dim3(1), dim3(1), 0, st, ps.regions, 0u, rp, 1u, (uint32_t *)nullptr);
    size_t ev = 0;
    const bool prof = ctx->profile;
    if (prof) (void)hipEventRecord(get_event(pp, ev++), st);  // (the two events that used to bracket ray generation)
    if (prof) (void)hipEventRecord(get_event(pp, ev++), st);
    const int grid = grid_fixed(ctx, n_slots), grid_shade = grid_fixed(ctx, n_slots, ctx->shade_blocks_per_cu);
    const uint32_t depth = ctx->cam.max_depth;
    // Rays expected at bounce b of this batch, from the most recent finished batch (0: unknown).
    auto expected = [&](uint32_t b) -> uint64_t {
        if (b == 0 || b >= ctx->bounce_profile.size() || ctx->bounce_profile[0] == 0u) return 0;
        return (uint64_t)((double)ctx->bounce_profile[b] * ((double)n_slots / (double)ctx->bounce_profile[0])) + 1u;
    };
    uint32_t bounces = 0;
    for (uint32_t b = 0; b < depth; b++, bounces++) {
        // one lane per expected ray, twice that for slack (any size is correct: the kernels loop over their queues)
        int g_ext = grid, g_sh = grid_shade;
        if (const uint64_t rays = expected(b)) {
            const int want = (int)std::min<uint64_t>((rays + BLOCK - 1) / BLOCK * 2u + 8u, 1u << 30);
            g_ext = std::min(grid, (want + 7) / 8 * 8);
            g_sh = std::min(grid_shade, (want + 7) / 8 * 8);
        }
        const FreshPaths fp{0u, b == 0u ? (uint32_t)n_slots : 0u};  // every path of the batch starts with its first step
        launch_extend(ctx, st, g_ext, ps, b, fp, pp.side[0], pp.fork, pp.join[0]);
        if (prof) (void)hipEventRecord(get_event(pp, ev++), st);
        if ((rc = launch_shades(ctx, pp, ShadeLaunch{ctx, st, &ps, g_sh, &bp, b, 0u, fp}))) return rc;
        if (ctx->nee) launch_shadow(ctx, st, g_ext, ps, b);
        if (prof) (void)hipEventRecord(get_event(pp, ev++), st);
    }
    if (paths_may_outlive_depth(ctx)) launch_finish(ctx, st, ps, bp, depth - 1u, 0u, 0u);
    if (bounces_out) *bounces_out = bounces;
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

// ---------------------------------------------------------------- rolling schedule
// Path state of a rolling pipe: radiance slots for n_regions batches of cap paths, and room for n_live rays in flight.
static int ensure_roll(pt_context *ctx, pt_context::Pipe &pp, pt_context::Roll &r, uint32_t n_regions, uint32_t cap, uint64_t n_live) {
    if (r.n_regions == n_regions && r.cap == cap && r.n_live == n_live) return PT_OK;
    (void)hipStreamSynchronize(pp.stream);  // the caller has finished the batches under way
    free_roll(ctx, r);
    const uint64_t total = (uint64_t)n_regions * cap;
    PathState ps{};
    int rc = alloc_path_state(ctx, r.allocs, pp.stream, ps, total, n_live, cap, n_regions);
    if (rc) return rc;
    ps.bounce_rays = nullptr;  // rays per bounce are only recorded where a batch has its launches to itself
    if ((rc = dev_alloc(ctx, r.allocs, &r.tick_rays, (size_t)pt_context::ROLL_RING))) return rc;
    ps.region_cap = cap;
    r.ps = ps;
    r.n_regions = n_regions;
    r.cap = cap;
    r.n_live = n_live;
    r.counted = r.tick;
    r.known_rays = 0;
    r.any_start = false;
    r.drain_rays.clear();
    return PT_OK;
}

// Bytes ensure_roll allocates per pipe, near enough.
static uint64_t roll_bytes(const pt_context *ctx, uint64_t n_regions, uint64_t cap, uint64_t n_live) {
    uint64_t unused, max_ray_chunks, max_chunks, max_shadow_chunks;
    queue_capacity(ctx, n_live, unused, max_ray_chunks);
    queue_capacity(ctx, n_live + cap, max_chunks, max_shadow_chunks);
    const uint64_t positions = max_ray_chunks * QUEUE_CHUNK;
    return n_regions * cap * 16u + positions * (2u * 52u + 20u) + cap * 20u + max_chunks * QUEUE_CHUNK * 4u * ctx->n_queues +
           (ctx->nee ? max_shadow_chunks * QUEUE_CHUNK * 52u : 0u);
}
// Rays in flight a rolling pipe has room for, given the paths started per launch.
static uint64_t roll_live(const pt_context *ctx, uint32_t depth, uint64_t cap) {
    const double f = std::min<double>(std::max(ctx->live_factor, 2.0), (double)depth);
    return (uint64_t)(f * (double)cap);
}

static void harvest_roll(pt_context *ctx, pt_context::Roll &r, int ring) {
    if (!r.prof_pending[ring]) return;
    r.prof_pending[ring] = false;
    if (hipEventSynchronize(r.prof[ring][3]) != hipSuccess) return;
    float ms = 0, since = 0;
    if (hipEventElapsedTime(&ms, r.prof[ring][0], r.prof[ring][1]) == hipSuccess) ctx->other_ms += ms;
    if (hipEventElapsedTime(&ms, r.prof[ring][1], r.prof[ring][2]) == hipSuccess) {
        ctx->extend_ms += ms;
        if (hipEventElapsedTime(&since, ctx->epoch, r.prof[ring][1]) == hipSuccess) ctx->extend_iv.push_back({since, since + ms});
    }
    if (hipEventElapsedTime(&ms, r.prof[ring][2], r.prof[ring][3]) == hipSuccess) ctx->shade_ms += ms;
    if (r.prof_walk[ring]) {
        if (hipEventElapsedTime(&ms, r.prof[ring][1], r.prof[ring][6]) == hipSuccess) {
            ctx->walk_ms += ms;
            ctx->walk_launches++;
        }
        if (r.prof_packet[ring] && hipEventElapsedTime(&ms, r.prof[ring][4], r.prof[ring][5]) == hipSuccess) {
            ctx->packet_ms += ms;
            ctx->packet_launches++;
        }
    }
    ctx->profiled_batches++;
}

// Takes note of the ticks that have finished since the last look (they finish in order).
static void roll_count_finished(pt_context::Roll &r) {
    while (r.counted < r.tick && hipEventQuery(r.done[r.counted % pt_context::ROLL_RING]) == hipSuccess) {
        r.known_rays = r.host_tick_rays[r.counted % pt_context::ROLL_RING];
        if (r.any_start && r.counted >= r.last_start) {
            const size_t j = (size_t)(r.counted - r.last_start);
            if (r.drain_rays.size() <= j) r.drain_rays.resize(j + 1, 0u);
            r.drain_rays[j] = r.known_rays;
        }
        r.counted++;
    }
}

// After a drain (every stream idle): from the survivors it counted tick by tick, the number of ticks after which
// so few paths of a batch are left that finishing them in one go (k_finish) costs less than keeping the batch's
// slots.  Returns 0 when nothing was learnt.
static uint32_t roll_learn(pt_context *ctx) {
    uint32_t best = 0;
    for (int p = 0; p < ctx->n_roll_pipes; p++) {
        pt_context::Roll &r = ctx->rolls[p];
        roll_count_finished(r);
        if (!r.any_start || r.counted != r.tick || r.cap == 0u || r.drain_rays.empty() || !r.due.empty()) continue;
        const uint32_t few = std::max<uint32_t>(8u, (uint32_t)((double)r.cap * 3.0e-7));
        uint32_t ticks = 0;
        for (size_t j = 0; j < r.drain_rays.size(); j++)
            if (r.drain_rays[j] <= few) {
                ticks = (uint32_t)j + 1u;
                break;
            }
        if (ticks == 0u) ticks = r.n_regions < ctx->cam.max_depth ? 2u * r.n_regions : ctx->cam.max_depth;  // more were left than the drain could show
        ticks = std::max(8u, (ticks + 3u) / 4u * 4u);
        best = std::max(best, std::min(ticks, ctx->cam.max_depth));
    }
    return best;
}

// One tick of the next pipe in turn.  `nb` (may be null): the batch to start with it; nb->n_slots <= cap.
// *started (may be null when nb is): whether the batch did start: it does not when its paths and the rays in
// flight might not fit the pipe's arrays, and the tick then only advances what is under way.
static int roll_step(pt_context *ctx, const BatchParams *nb, bool *started = nullptr) {
    const int p = (int)(ctx->roll_seq % (uint64_t)ctx->n_roll_pipes);
    HIP_TRY(ctx, pipe_init(ctx, p));
    pt_context::Pipe &pp = ctx->pipes[p];
    pt_context::Roll &r = ctx->rolls[p];
    const int ring = (int)(r.tick % pt_context::ROLL_RING);
    auto wait_for = [&](uint64_t tick) -> int {
        auto w0 = std::chrono::steady_clock::now();
        HIP_TRY(ctx, hipEventSynchronize(r.done[tick % pt_context::ROLL_RING]));
        ctx->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        return PT_OK;
    };
    auto count_finished = [&]() { roll_count_finished(r); };
    if (r.tick >= (uint64_t)pt_context::ROLL_RING) {  // this tick reuses the ring entry of the tick ROLL_RING before it
        if (int rc = wait_for(r.tick)) return rc;
        harvest_roll(ctx, r, ring);
    }
    count_finished();
    if (nb) {
        // rays queued for this tick <= what the last finished tick queued + the paths started since
        for (;;) {
            uint64_t in_flight = r.known_rays;
            for (uint64_t t = r.counted; t < r.tick; t++) in_flight += r.ring_new[t % pt_context::ROLL_RING];
            if (in_flight + nb->n_slots <= r.n_live) break;
            if (r.counted == r.tick) {  // everything is known and there is no room: this tick starts nothing
                nb = nullptr;
                break;
            }
            if (int rc = wait_for(r.counted)) return rc;
            count_finished();
        }
    }
    if (started) *started = nb != nullptr;
    hipStream_t st = pp.stream;
    r.ps.refill_below = ctx->refill_below;
    r.ps.leaf_batch = ctx->leaf_batch;
    r.ps.next_rays = r.tick_rays + ring;
    // first hits arrive in path order: their queues are best dealt out in advance; the vertices of later bounces are not
    // (nor are they where many material types share a step: 11 queues on three streams lose 12 % of the shading time to it)
    r.ps.shade_claim = ctx->shade_claim >= 0 ? (uint32_t)ctx->shade_claim : ((nb && (uint64_t)nb->n_slots >= r.known_rays) || ctx->n_queues > 4u ? 0u : 4u);
    r.ps.shade_claim_from = ctx->shade_claim >= 0 ? 0u : 2048u;
    const PathState ps = r.ps;
    const uint32_t depth = ctx->cam.max_depth;
    const uint32_t region = (uint32_t)(r.tick % r.n_regions), base = region * r.cap;
    const uint32_t parity = (uint32_t)(r.tick & 1u);
    const bool prof = ctx->profile;
    if (prof) (void)hipEventRecord(r.prof[ring][0], st);
    FreshPaths fp{base, 0u};
    RegionParams rp{};
    if (nb) {
        fp.n_new = nb->n_slots;
        rp.seed_mixed = nb->seed_mixed;
        rp.stratum_first = nb->stratum_first;
        rp.n_strata = nb->n_strata;
        r.due.push_back({r.tick + r.n_regions - 1u, region, nb->n_strata});
        r.last_start = r.tick;
        r.any_start = true;
        r.drain_rays.clear();
    }
    r.ring_new[ring] = fp.n_new;
    hipLaunchKernelGGL(k_tick_begin, dim3(1), dim3(1), 0, st, ps.regions, region, rp, nb ? 1u : 0u, ps.next_rays);
    if (prof) (void)hipEventRecord(r.prof[ring][1], st);
    const uint64_t bound = r.n_live + r.cap;
    r.prof_walk[ring] = prof && ctx->use_wide() && !ctx->counters;
    launch_extend(ctx, st, grid_fixed(ctx, bound), ps, parity, fp, pp.side[0], pp.fork, pp.join[0], r.known_rays, r.prof_walk[ring] ? &r.prof[ring][4] : nullptr,
                  &r.prof_packet[ring]);
    if (prof) (void)hipEventRecord(r.prof[ring][2], st);
    BatchParams common{};
    common.npix = ctx->cam.width * ctx->cam.height;
    common.bsdf_prob = ctx->nee ? 1.0f : ctx->bsdf_prob;
    common.background = ctx->cam.background;
    const int grid_shade = grid_fixed(ctx, bound, ctx->shade_blocks_per_cu);
    if (int rc = launch_shades(ctx, pp, ShadeLaunch{ctx, st, &ps, grid_shade, &common, parity, 0u, fp})) return rc;
    if (ctx->nee) launch_shadow(ctx, st, grid_fixed(ctx, bound), ps, parity);
    HIP_TRY(ctx, hipMemcpyAsync(r.host_tick_rays + ring, ps.next_rays, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    if (prof) {
        (void)hipEventRecord(r.prof[ring][3], st);
        r.prof_pending[ring] = true;
    }
    if (!r.due.empty() && r.due.front().tick == r.tick) {
        // in stratum order over all pipes: ticks go round the pipes, so batches fall due in the order they started
        const pt_context::Due d = r.due.front();
        r.due.pop_front();
        if (paths_may_outlive_depth(ctx) || r.n_regions < depth) launch_finish(ctx, st, ps, common, parity, d.region, (uint32_t)(r.finishes++ & 1u));
        if (ctx->last_accumulated) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->last_accumulated, 0));
        hipLaunchKernelGGL(k_accumulate, dim3(grid_for(ctx, common.npix)), dim3(BLOCK), 0, st, ps.contrib + (size_t)d.region * r.cap, ctx->accum, common.npix,
                           d.n_strata);
        HIP_TRY(ctx, hipEventRecord(r.accumulated[ring], st));
        ctx->last_accumulated = r.accumulated[ring];
    }
    HIP_TRY(ctx, hipEventRecord(r.done[ring], st));
    r.tick++;
    ctx->roll_seq++;
    return PT_OK;
}

// Enqueues the ticks that the batches under way still need.
static int roll_flush(pt_context *ctx) {
    for (;;) {
        bool any = false;
        for (int p = 0; p < ctx->n_roll_pipes; p++) any = any || !ctx->rolls[p].due.empty();
        if (!any) return PT_OK;
        if (int rc = roll_step(ctx, nullptr)) return rc;
    }
}

// Strata per tick of the rolling schedule for the current camera, 0 when the batch-at-a-time schedule is
// to be used (switched off, counting or reference-order traversal, or the regions would not fit).
static uint32_t roll_strata_now(pt_context *ctx);
static uint32_t roll_strata(pt_context *ctx) {
    // decided once per scene / camera / option change: the answer depends on the free memory at that time, and
    // a different one later would mean reallocating the path state in the middle of a render
    if (ctx->roll_k < 0) {
        ctx->roll_k = (int64_t)roll_strata_now(ctx);
        ctx->roll_seen = 0;
    }
    return (uint32_t)ctx->roll_k;
}
static uint32_t roll_strata_now(pt_context *ctx) {
    const uint32_t depth = ctx->cam.max_depth;
    if (ctx->roll_paths == 0 || ctx->counters || !ctx->use_wide() || depth == 0) return 0;
    const uint64_t npix = (uint64_t)ctx->cam.width * ctx->cam.height;
    uint64_t k = std::max<uint64_t>(1, std::min(ctx->roll_paths, ctx->max_paths) / npix);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    uint64_t have = free_b;
    for (int p = 0; p < ctx->n_roll_pipes; p++)
        for (const auto &a : ctx->rolls[p].allocs) have += a.second;  // what is there would be replaced
    // (while the number of pipes follows the launch size, the budget is that of two: small launches get a second pipe)
    const uint64_t budget = have / 2 / (uint64_t)(ctx->roll_pipes_auto ? 2 : ctx->n_roll_pipes);
    auto bytes = [&](uint64_t kk) { return roll_bytes(ctx, ctx->roll_regions(), npix * kk, roll_live(ctx, depth, npix * kk)); };
    // slots are 32-bit numbers, and every region of a pipe has them
    auto too_big = [&](uint64_t kk) { return bytes(kk) > budget || (uint64_t)ctx->roll_regions() * npix * kk > (1ull << 31); };
    while (k > 1 && too_big(k)) k--;
    if (too_big(k)) return 0;
    return (uint32_t)k;
}

static int check_render_args(pt_context *ctx, uint32_t b, uint32_t e) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_scene || !ctx->have_camera) return fail(ctx, PT_ERR_STATE, "upload a scene and set a camera before rendering");
    const uint64_t total = (uint64_t)ctx->cam.sqrt_spp * ctx->cam.sqrt_spp;
    if (b > e || e > total) return fail(ctx, PT_ERR_INVALID, "stratum range [%u,%u) outside [0,%llu)", b, e, (unsigned long long)total);
    // LightBuffer::sample draws gen_range(0..lights.len()) (src/objects.rs:236-237): rand panics on an empty range
    if (!ctx->nee && ctx->bsdf_prob < 1.0f && ctx->scene.n_light_total == 0)
        return fail(ctx, PT_ERR_INVALID, "BSDF_SAMPLE_PROBABILITY below 1 needs a light to sample, and the scene has none");
    if (ctx->nee && (ctx->counters || !ctx->use_wide()))
        return fail(ctx, PT_ERR_UNSUPPORTED, "next-event estimation runs on the ray-drawing traversal only (no test counters, no reference-order traversal)");
    return PT_OK;
}

int pt_render(pt_context *ctx, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed) {
    int rc = check_render_args(ctx, stratum_begin, stratum_end);
    if (rc) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t npix = ctx->cam.width * ctx->cam.height;
    // Batches as large as max_paths allows and of equal size: the bigger the batch, the smaller
    // the share of its thin late bounces; consecutive calls keep the other pipes busy.
    const uint32_t total = stratum_end - stratum_begin;
    if (total == 0) return PT_OK;
    if (ctx->cam.max_depth == 0) {
        // ray_color returns zeros before it traces anything (src/camera.rs:271-273): the passes count, and add nothing
        ctx->n_samples += total;
        ctx->untraced_paths += (uint64_t)total * npix;
        return PT_OK;
    }
    if (uint32_t k_tick = roll_strata(ctx)) {
        // the path state is sized by the work asked for: a caller that renders a few strata at a time gets launches
        // (and slot arrays) of that size, not of the largest size a launch may have; a larger request later grows them
        ctx->roll_seen = std::max(ctx->roll_seen, std::min(total, k_tick));
        k_tick = std::min(k_tick, ctx->roll_seen);
        const uint32_t depth = ctx->cam.max_depth, cap_slots = npix * k_tick;
        const uint64_t n_live = roll_live(ctx, depth, cap_slots);
        if (ctx->roll_pipes_auto) {
            const int want = cap_slots <= pt_context::ROLL_SMALL_LAUNCH ? 2 : 1;
            if (want != ctx->n_roll_pipes) {
                if ((rc = sync_all(ctx))) return rc;  // what is under way is finished by the pipes that started it
                for (int p = want; p < ctx->n_roll_pipes; p++) free_roll(ctx, ctx->rolls[p]);
                ctx->n_roll_pipes = want;
            }
        }
        bool fits = true;
        const uint32_t regions = ctx->roll_regions();
        for (int p = 0; p < ctx->n_roll_pipes; p++) fits = fits && ctx->rolls[p].n_regions == regions && ctx->rolls[p].cap == cap_slots && ctx->rolls[p].n_live == n_live;
        if (!fits) {
            sync_pipes(ctx);
            for (int p = 0; p < ctx->n_roll_pipes; p++) {
                HIP_TRY(ctx, pipe_init(ctx, p));
                if ((rc = ensure_roll(ctx, ctx->pipes[p], ctx->rolls[p], regions, cap_slots, n_live))) return rc;
            }
        }
        const uint32_t n_batches = (total + k_tick - 1) / k_tick, per_batch = (total + n_batches - 1) / n_batches;
        for (uint32_t s = stratum_begin; s < stratum_end;) {
            const uint32_t k = std::min(per_batch, stratum_end - s);
            BatchParams bp{mix64_host(seed), s, k, npix, npix * k, ctx->bsdf_prob, ctx->cam.background};
            for (bool started = false; !started;)  // ticks that cannot take the batch yet still advance the paths under way
                if ((rc = roll_step(ctx, &bp, &started))) return rc;
            ctx->n_samples += k;
            s += k;
        }
        HIP_TRY(ctx, hipGetLastError());
        ctx->render_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return PT_OK;
    }
    if ((rc = roll_flush(ctx))) return rc;  // what the rolling schedule has under way is accumulated first
    const uint32_t cap = (uint32_t)std::max<uint64_t>(1, ctx->max_paths / npix);
    const uint32_t n_batches = (total + cap - 1) / cap;
    const uint32_t per_batch = (total + n_batches - 1) / n_batches;
    // k_accumulate of batch i runs on batch i's stream after batch i-1's k_accumulate, so each
    // pixel receives its samples in stratum order whatever the overlap (camera.rs:212-217).
    for (uint32_t s = stratum_begin; s < stratum_end;) {
        HIP_TRY(ctx, pipe_init(ctx, (int)(ctx->batch_seq % (uint64_t)ctx->n_pipes)));
        pt_context::Pipe &pp = ctx->pipes[ctx->batch_seq % (uint64_t)ctx->n_pipes];
        const uint32_t k = std::min(per_batch, stratum_end - s);
        // the host runs at most two batches ahead of each pipe
        hipEvent_t done = pp.accumulated[pp.batches & 1u];
        if (pp.batches >= 2) {
            auto w0 = std::chrono::steady_clock::now();
            HIP_TRY(ctx, hipEventSynchronize(done));
            ctx->host_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
            const uint32_t have = pp.host_chunks_depth[pp.batches & 1u];
            if (have) {
                ctx->bounce_profile.assign(pp.host_chunks[pp.batches & 1u], pp.host_chunks[pp.batches & 1u] + have);
                ctx->bounce_profile[0] = pp.host_paths[pp.batches & 1u];  // bounce 0 traces every path of the batch
            }
        }
        uint32_t bounces = 0;
        if ((rc = run_batch(ctx, pp, s, k, seed, &bounces))) return rc;
        if (ctx->last_accumulated) HIP_TRY(ctx, hipStreamWaitEvent(pp.stream, ctx->last_accumulated, 0));
        const size_t ev = 2 + 2 * (size_t)bounces;
        if (ctx->profile) (void)hipEventRecord(get_event(pp, ev), pp.stream);
        hipLaunchKernelGGL(k_accumulate, dim3(grid_for(ctx, npix)), dim3(BLOCK), 0, pp.stream, pp.ps.contrib, ctx->accum, npix, k);
        if (ctx->profile) {
            (void)hipEventRecord(get_event(pp, ev + 1), pp.stream);
            pp.events_pending[pp.batches & 1u] = bounces;
        }
        {
            const uint32_t n_b = std::min<uint32_t>(ctx->cam.max_depth, BOUNCE_PROFILE_LEN);
            HIP_TRY(ctx, hipMemcpyAsync(pp.host_chunks[pp.batches & 1u], pp.ps.bounce_rays, n_b * sizeof(uint32_t), hipMemcpyDeviceToHost, pp.stream));
            pp.host_chunks_depth[pp.batches & 1u] = n_b;
            pp.host_paths[pp.batches & 1u] = (uint32_t)((uint64_t)npix * k);
        }
        HIP_TRY(ctx, hipEventRecord(done, pp.stream));
        ctx->last_accumulated = done;
        pp.batches++;
        ctx->batch_seq++;
        ctx->n_samples += k;  // `samples += 1` per pass (camera.rs:199)
        s += k;
    }
    HIP_TRY(ctx, hipGetLastError());
    ctx->render_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PT_OK;
}

int pt_render_samples(pt_context *ctx, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed, float *out) {
    int rc = check_render_args(ctx, stratum_begin, stratum_end);
    if (rc) return rc;
    if (!out) return fail(ctx, PT_ERR_INVALID, "out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = ctx->cam.width * ctx->cam.height;
    const uint32_t per_batch = (uint32_t)std::max<uint64_t>(1, ctx->max_paths / npix);
    pt_context::Pipe &pp = ctx->pipes[0];
    const bool prof = ctx->profile;
    ctx->profile = false;
    std::vector<float4> host;
    if (ctx->cam.max_depth == 0) {  // as in pt_render: nothing is traced
        std::memset(out, 0, (size_t)(stratum_end - stratum_begin) * npix * 3 * sizeof(float));
        ctx->untraced_paths += (uint64_t)(stratum_end - stratum_begin) * npix;
        ctx->profile = prof;
        return PT_OK;
    }
    for (uint32_t s = stratum_begin; s < stratum_end;) {
        const uint32_t k = std::min(per_batch, stratum_end - s);
        if ((rc = run_batch(ctx, pp, s, k, seed))) break;
        const size_t n = (size_t)npix * k;
        host.resize(n);
        hipError_t e = hipMemcpyAsync(host.data(), pp.ps.contrib, n * sizeof(float4), hipMemcpyDeviceToHost, pp.stream);
        if (e == hipSuccess) e = hipMemsetAsync(pp.ps.contrib, 0, n * sizeof(float4), pp.stream);  // as k_accumulate leaves them
        if (e == hipSuccess) e = hipStreamSynchronize(pp.stream);
        if (e != hipSuccess) {
            rc = fail(ctx, PT_ERR_HIP, "pt_render_samples: %s", hipGetErrorString(e));
            break;
        }
        for (size_t i = 0; i < n; i++) {  // slot = pixel * k + j  ->  out[stratum][pixel]
            const size_t pixel = i / k, j = i % k;
            float *o = out + ((size_t)(s - stratum_begin + j) * npix + pixel) * 3;
            o[0] = host[i].x;
            o[1] = host[i].y;
            o[2] = host[i].z;
        }
        s += k;
    }
    ctx->profile = prof;
    return rc;
}

int pt_generate_rays(pt_context *ctx, uint32_t stratum, uint64_t seed, float *rays) {
    int rc = check_render_args(ctx, stratum, stratum + 1);
    if (rc) return rc;
    if (!rays) return fail(ctx, PT_ERR_INVALID, "rays is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = ctx->cam.width * ctx->cam.height;
    float *d = nullptr;
    Staging stg(ctx);
    if ((rc = stg.get(0, (size_t)npix * 8, &d))) return rc;
    BatchParams bp{mix64_host(seed), stratum, 1u, npix, npix, ctx->bsdf_prob};
    hipLaunchKernelGGL(k_raygen_only, dim3(grid_for(ctx, npix)), dim3(BLOCK), 0, ctx->stream, ctx->cam, bp, d);
    hipError_t e = hipMemcpyAsync(rays, d, (size_t)npix * 8 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_generate_rays: %s", hipGetErrorString(e));
    return PT_OK;
}

// PrimitiveBuffer::hit for caller-supplied rays; with_moving: then the moving spheres at times[i] (null: time 0).
static int trace_first_hit(pt_context *ctx, const float *rays, const float *times, bool with_moving, uint64_t n, pt_hit *out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "upload a scene before tracing");
    if (n == 0) return PT_OK;
    if (!rays || !out) return fail(ctx, PT_ERR_INVALID, "NULL ray or output array");
    if (n > (1ull << 30)) return fail(ctx, PT_ERR_INVALID, "too many rays in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pt_context::Pipe &pp = ctx->pipes[0];
    int rc = ensure_paths(ctx, pp, n);
    if (rc) return rc;
    with_moving = with_moving && ctx->scene.n_moving != 0;
    float *d_rays = nullptr, *d_times = nullptr;
    pt_hit *d_out = nullptr;
    Staging stg(ctx);
    if ((rc = stg.get(0, n * 8, &d_rays)) || (rc = stg.get(1, n, &d_out))) return rc;
    if (with_moving && times && (rc = stg.get(2, n, &d_times))) return rc;
    hipError_t e;
    hipStream_t st = ctx->stream;
    e = hipMemcpyAsync(d_rays, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && d_times) e = hipMemcpyAsync(d_times, times, n * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_load_rays, dim3(grid_for(ctx, n)), dim3(BLOCK), 0, st, d_rays, pp.ps, (uint32_t)n);
        const dim3 g(grid_fixed(ctx, n)), b(BLOCK);
        const bool insts = ctx->scene.world.insts.prim_count != 0;
        if (ctx->profile) (void)hipEventRecord(ctx->rolls[0].prof[0][0], st);  // the search itself, for pt_stats::extend_ms
        if (ctx->trace_packet && ctx->packet_ok) {
            if (insts) hipLaunchKernelGGL((k_trace_packet<true>), g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n);
            else hipLaunchKernelGGL((k_trace_packet<false>), g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n);
        } else if (ctx->use_wide()) {
            if (insts) hipLaunchKernelGGL((k_trace_walk<true>), g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n, ctx->stats);
            else if (walk_tail_applies(ctx->scene.world)) hipLaunchKernelGGL((k_trace_walk<false, true>), g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n, ctx->stats);
            else hipLaunchKernelGGL((k_trace_walk<false>), g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n, ctx->stats);
        } else {
            hipLaunchKernelGGL(k_trace, g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n);
            if (insts) hipLaunchKernelGGL(k_trace_inst, g, b, 0, st, ctx->scene, pp.ps, (uint32_t)n);
        }
        if (with_moving) hipLaunchKernelGGL(k_trace_moving, dim3(grid_for(ctx, n)), b, 0, st, ctx->scene, pp.ps, (uint32_t)n, d_times);
        if (ctx->profile) (void)hipEventRecord(ctx->rolls[0].prof[0][1], st);
        hipLaunchKernelGGL(k_trace_out, dim3(grid_for(ctx, n)), b, 0, st, ctx->scene, pp.ps, (uint32_t)n, d_out);
        e = hipMemcpyAsync(out, d_out, n * sizeof(pt_hit), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_trace_first_hit: %s", hipGetErrorString(e));
    if (ctx->profile) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->rolls[0].prof[0][0], ctx->rolls[0].prof[0][1]) == hipSuccess) ctx->extend_ms += ms;
    }
    return sync_all(ctx);
}
int pt_trace_first_hit(pt_context *ctx, const float *rays, uint64_t n, pt_hit *out) { return trace_first_hit(ctx, rays, nullptr, false, n, out); }
int pt_trace_first_hit_at(pt_context *ctx, const float *rays, const float *times, uint64_t n, pt_hit *out) {
    return trace_first_hit(ctx, rays, times, true, n, out);
}

// ---------------------------------------------------------------- first hits without allocating
// pt_trace_first_hit_static: the closest-hit query for callers that must not allocate.  What
// pt_trace_first_hit takes from the context's path state (rays, hits, the ray queue, its control
// words, the part of the traversal stacks that does not fit LDS) lives in one file-scope device
// array instead.  Every GPU holds its own copy of that array at its own address: the address is
// looked up once per device, with that device current, and from then on only that address is used
// on the host side (the array's name is not a device address in host code).
constexpr uint32_t STATIC_MAX_RAYS = PT_STATIC_TRACE_MAX_RAYS;
constexpr uint32_t STATIC_BLOCKS = 128;                       // workgroups of its launches: the stack overflow is [level][lane of the launch]
constexpr uint32_t STATIC_STACK_LEVELS = 2 * 64 + 4;          // deepest stack a scene that passes nodes_ok can need (world + instanced set)
constexpr uint32_t STATIC_EXTRA_LEVELS = STATIC_STACK_LEVELS > (uint32_t)LDS_LEVELS ? STATIC_STACK_LEVELS - LDS_LEVELS : 0u;
constexpr size_t STATIC_QUEUE_ENTRIES = ((size_t)STATIC_MAX_RAYS + QUEUE_CHUNK - 1) / QUEUE_CHUNK * QUEUE_CHUNK;
constexpr size_t STATIC_OFF_RAY_O = 0;
constexpr size_t STATIC_OFF_RAY_D = STATIC_OFF_RAY_O + (size_t)STATIC_MAX_RAYS * sizeof(float4);
constexpr size_t STATIC_OFF_HIT = STATIC_OFF_RAY_D + (size_t)STATIC_MAX_RAYS * sizeof(float4);
constexpr size_t STATIC_OFF_HIT2 = STATIC_OFF_HIT + (size_t)STATIC_MAX_RAYS * sizeof(float4);
constexpr size_t STATIC_OFF_QUEUE = STATIC_OFF_HIT2 + (size_t)STATIC_MAX_RAYS * sizeof(uint32_t);
constexpr size_t STATIC_OFF_CTL = STATIC_OFF_QUEUE + STATIC_QUEUE_ENTRIES * sizeof(uint32_t);
constexpr size_t STATIC_OFF_OVERFLOW = (STATIC_OFF_CTL + (size_t)CTL_WORDS * sizeof(uint32_t) + 255) / 256 * 256;
constexpr size_t STATIC_BYTES = STATIC_OFF_OVERFLOW + (size_t)STATIC_EXTRA_LEVELS * STATIC_BLOCKS * BLOCK * sizeof(uint2);
static __device__ __align__(256) unsigned char scratch[STATIC_BYTES];

constexpr int STATIC_MAX_DEVICES = 64;
static unsigned char *g_scratch_ptr[STATIC_MAX_DEVICES];  // device address of `scratch` on each GPU seen so far
static std::mutex g_scratch_mutex[STATIC_MAX_DEVICES];    // held for the length of a call that uses it

// Called by pt_create with `device` current.
static hipError_t scratch_lookup(int device) {
    if (device < 0 || device >= STATIC_MAX_DEVICES) return hipSuccess;  // such a device has no static first-hit query
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = (unsigned char *)p;
    return hipSuccess;
}

int pt_trace_first_hit_static(pt_context *ctx, const float *rays_device, uint64_t n, pt_hit *out_device) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "upload a scene before tracing");
    if (n == 0) return PT_OK;
    if (!rays_device || !out_device) return fail(ctx, PT_ERR_INVALID, "NULL ray or output array");
    if (n > STATIC_MAX_RAYS)
        return fail(ctx, PT_ERR_INVALID, "pt_trace_first_hit_static takes at most %u rays per call (got %llu): split the call, or use pt_trace_first_hit",
                    STATIC_MAX_RAYS, (unsigned long long)n);
    if (ctx->stack_need > STATIC_STACK_LEVELS) return fail(ctx, PT_ERR_UNSUPPORTED, "the scene's BVHs are deeper than the static traversal stacks");
    unsigned char *base = ctx->device < STATIC_MAX_DEVICES ? g_scratch_ptr[ctx->device] : nullptr;
    if (!base) return fail(ctx, PT_ERR_STATE, "no static scratch on device %d", ctx->device);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PathState ps{};
    ps.ray_o[0] = (float4 *)(base + STATIC_OFF_RAY_O);
    ps.ray_d[0] = (float4 *)(base + STATIC_OFF_RAY_D);
    ps.hit = (float4 *)(base + STATIC_OFF_HIT);
    ps.hit2 = (uint32_t *)(base + STATIC_OFF_HIT2);
    ps.ray_queue[0] = (uint32_t *)(base + STATIC_OFF_QUEUE);
    ps.ctl = (uint32_t *)(base + STATIC_OFF_CTL);
    ps.overflow = (uint2 *)(base + STATIC_OFF_OVERFLOW);
    ps.overflow_stride = STATIC_BLOCKS * BLOCK;
    ps.max_chunks = ps.max_ray_chunks = (uint32_t)(STATIC_QUEUE_ENTRIES / QUEUE_CHUNK);
    ps.queue_overflow = &ctx->stats->queue_overflow;
    ps.n_slots = (uint32_t)n;
    ps.refill_below = ctx->refill_below;
    ps.leaf_batch = ctx->leaf_batch;
    ps.shade_claim = 0u;
    ps.shade_claim_from = 0u;
    hipStream_t st = ctx->stream;
    // every context of this GPU uses the same array: one call at a time per device (a mutex allocates nothing)
    std::lock_guard<std::mutex> one_at_a_time(g_scratch_mutex[ctx->device]);
    // the array keeps what the last call on this GPU left in it: the control words start from zero
    HIP_TRY(ctx, hipMemsetAsync(ps.ctl, 0, (size_t)CTL_WORDS * sizeof(uint32_t), st));
    const uint64_t want = (n + BLOCK - 1) / BLOCK;
    const dim3 g((uint32_t)std::min<uint64_t>(want, STATIC_BLOCKS)), b(BLOCK);
    hipLaunchKernelGGL(k_load_rays, g, b, 0, st, rays_device, ps, (uint32_t)n);
    const bool insts = ctx->scene.world.insts.prim_count != 0;
    if (ctx->use_wide()) {
        if (insts) hipLaunchKernelGGL((k_trace_walk<true>), g, b, 0, st, ctx->scene, ps, (uint32_t)n, ctx->stats);
        else if (walk_tail_applies(ctx->scene.world)) hipLaunchKernelGGL((k_trace_walk<false, true>), g, b, 0, st, ctx->scene, ps, (uint32_t)n, ctx->stats);
        else hipLaunchKernelGGL((k_trace_walk<false>), g, b, 0, st, ctx->scene, ps, (uint32_t)n, ctx->stats);
    } else {
        hipLaunchKernelGGL(k_trace, g, b, 0, st, ctx->scene, ps, (uint32_t)n);
        if (insts) hipLaunchKernelGGL(k_trace_inst, g, b, 0, st, ctx->scene, ps, (uint32_t)n);
    }
    hipLaunchKernelGGL(k_trace_out, g, b, 0, st, ctx->scene, ps, (uint32_t)n, out_device);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipGetLastError());
    return PT_OK;
}

int pt_light_pdf(pt_context *ctx, const float *rays, uint64_t n, int tris_by_solid_angle, float *out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "upload a scene before asking for light pdfs");
    if (n == 0) return PT_OK;
    if (!rays || !out) return fail(ctx, PT_ERR_INVALID, "NULL ray or output array");
    if (n > (1ull << 30)) return fail(ctx, PT_ERR_INVALID, "too many rays in one call");
    if (ctx->scene.n_light_total == 0) return fail(ctx, PT_ERR_STATE, "the scene has no lights");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_rays = nullptr, *d_out = nullptr;
    Staging stg(ctx);
    int rc;
    if ((rc = stg.get(0, n * 8, &d_rays)) || (rc = stg.get(1, n, &d_out))) return rc;
    hipError_t e;
    hipStream_t st = ctx->stream;
    e = hipMemcpyAsync(d_rays, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_light_pdf, dim3(grid_for(ctx, n)), dim3(BLOCK), 0, st, ctx->scene, d_rays, (uint32_t)n, tris_by_solid_angle, d_out);
        e = hipMemcpyAsync(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_light_pdf: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_material_probe(pt_context *ctx, int32_t material, uint64_t seed, const pt_material_probe_in *in, uint64_t n, pt_material_probe_out *out) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "upload a scene before probing its materials");
    if (material < 0 || (uint32_t)material >= ctx->scene.n_materials) return fail(ctx, PT_ERR_INVALID, "material %d out of range (the scene has %u)", material, ctx->scene.n_materials);
    if (n == 0) return PT_OK;
    if (!in || !out) return fail(ctx, PT_ERR_INVALID, "NULL probe or output array");
    if (n > (1ull << 28)) return fail(ctx, PT_ERR_INVALID, "too many probes in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pt_material_probe_in *d_in = nullptr;
    pt_material_probe_out *d_out = nullptr;
    Staging stg(ctx);
    int rc;
    if ((rc = stg.get(0, n, &d_in)) || (rc = stg.get(1, n, &d_out))) return rc;
    hipError_t e;
    hipStream_t st = ctx->stream;
    e = hipMemcpyAsync(d_in, in, n * sizeof *d_in, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_material_probe, dim3(grid_for(ctx, n)), dim3(BLOCK), 0, st, ctx->scene, (uint32_t)material, mix64_host(seed), d_in, (uint32_t)n, d_out);
        e = hipMemcpyAsync(out, d_out, n * sizeof *d_out, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, PT_ERR_HIP, "pt_material_probe: %s", hipGetErrorString(e));
    return PT_OK;
}

int pt_read_accum(pt_context *ctx, float *rgb_sum, uint32_t *n_samples) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    const size_t npix = (size_t)ctx->cam.width * ctx->cam.height;
    if (rgb_sum) HIP_TRY(ctx, hipMemcpy(rgb_sum, ctx->accum, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (n_samples) *n_samples = ctx->n_samples;
    return PT_OK;
}

float *ptc_accum(pt_context *ctx) { return ctx->have_camera ? ctx->accum : nullptr; }
uint64_t ptc_npix(const pt_context *ctx) { return ctx->have_camera ? (uint64_t)ctx->cam.width * ctx->cam.height : 0; }
uint32_t ptc_n_samples(const pt_context *ctx) { return ctx->n_samples; }
hipStream_t ptc_stream(pt_context *ctx) { return ctx->stream; }
int ptc_device(const pt_context *ctx) { return ctx->device; }
int ptc_tonemap(pt_context *ctx, const float *accum_device, uint32_t n_samples, uint8_t *rgb_host) {
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t npix = ctx->cam.width * ctx->cam.height;
    hipLaunchKernelGGL(k_tonemap, dim3(grid_for(ctx, npix)), dim3(BLOCK), 0, ctx->stream, accum_device, ctx->rgb8, npix, (float)n_samples);
    HIP_TRY(ctx, hipMemcpyAsync(rgb_host, ctx->rgb8, (size_t)npix * 3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_read_rgb8(pt_context *ctx, uint8_t *rgb) {
    if (!ctx || !rgb) return fail(ctx, PT_ERR_INVALID, "NULL argument");
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    return ptc_tonemap(ctx, ctx->accum, ctx->n_samples, rgb);
}

int pt_read_triangle_order(pt_context *ctx, uint32_t *order, uint32_t n) {
    if (!ctx || !order) return fail(ctx, PT_ERR_INVALID, "pt_read_triangle_order: NULL argument");
    if (!ctx->have_scene) return fail(ctx, PT_ERR_STATE, "no scene uploaded");
    if (n != ctx->scene.n_tris) return fail(ctx, PT_ERR_INVALID, "the scene has %u triangles, not %u", ctx->scene.n_tris, n);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n) HIP_TRY(ctx, hipMemcpy(order, ctx->scene.tri_order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_reset_accum(pt_context *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    HIP_TRY(ctx, hipMemset(ctx->accum, 0, (size_t)ctx->cam.width * ctx->cam.height * 3 * sizeof(float)));
    ctx->n_samples = 0;
    return PT_OK;
}

int pt_copy_accum_to_device(pt_context *ctx, void *dst_device) {
    if (!ctx || !dst_device) return fail(ctx, PT_ERR_INVALID, "NULL argument");
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dst_device, ctx->accum, (size_t)ctx->cam.width * ctx->cam.height * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_set_accum_from_device(pt_context *ctx, const void *src_device, uint32_t n_samples) {
    if (!ctx || !src_device) return fail(ctx, PT_ERR_INVALID, "NULL argument");
    if (!ctx->have_camera) return fail(ctx, PT_ERR_STATE, "no camera set");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->accum, src_device, (size_t)ctx->cam.width * ctx->cam.height * 3 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->n_samples = n_samples;
    return PT_OK;
}

int pt_stats_get(pt_context *ctx, pt_stats *out) {
    if (!ctx || !out) return fail(ctx, PT_ERR_INVALID, "NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    DevStats ds;
    HIP_TRY(ctx, hipMemcpy(&ds, ctx->stats, sizeof ds, hipMemcpyDeviceToHost));
    std::memset(out, 0, sizeof *out);
    out->rays = ds.rays_total;
    out->paths = ds.paths_total + ctx->untraced_paths;
    out->extend_launches = ds.extend_nonempty;
    out->extend_ms = ctx->extend_ms;
    out->extend_busy_ms = ctx->extend_busy_ms;
    out->shade_ms = ctx->shade_ms;
    out->other_ms = ctx->other_ms;
    out->render_ms = ctx->render_ms;
    out->host_wait_ms = ctx->host_wait_ms;
    out->node_tests = ds.node_tests;
    out->tri_tests = ds.tri_tests;
    out->sphere_tests = ds.sphere_tests;
    out->quad_tests = ds.quad_tests;
    out->bad_refs = ds.bad_refs[0] + ds.bad_refs[1];
    out->walk_ms = ctx->walk_ms;
    out->packet_ms = ctx->packet_ms;
    out->walk_launches = ctx->walk_launches;
    out->shadow_rays = ds.shadow_rays;
    out->packet_launches = ctx->packet_launches;
    out->device_bytes = ctx->device_bytes;
    if (getenv("PT_DEBUG_BOUNCES") && ctx->profiled_batches) {  // per-bounce means of the event timings, for tuning
        fprintf(stderr, "bounce: extend ms, shade ms (mean of %llu batches), rays in the last profile\n", (unsigned long long)ctx->profiled_batches);
        for (size_t b = 0; b < ctx->bounce_extend_ms.size() && b < 12; b++)
            fprintf(stderr, "%3zu: %8.3f %8.3f %10u\n", b, ctx->bounce_extend_ms[b] / ctx->profiled_batches, ctx->bounce_shade_ms[b] / ctx->profiled_batches,
                    b < ctx->bounce_profile.size() ? ctx->bounce_profile[b] : 0u);
        double te = 0, ts = 0;
        for (size_t b = 12; b < ctx->bounce_extend_ms.size(); b++) te += ctx->bounce_extend_ms[b], ts += ctx->bounce_shade_ms[b];
        fprintf(stderr, "12+: %8.3f %8.3f\n", te / ctx->profiled_batches, ts / ctx->profiled_batches);
    }
    return PT_OK;
}

int pt_stats_reset(pt_context *ctx) {
    if (!ctx) return PT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = restart_epoch(ctx)) return rc;
    HIP_TRY(ctx, hipMemset(ctx->stats, 0, sizeof(DevStats)));
    ctx->extend_ms = ctx->extend_busy_ms = ctx->shade_ms = ctx->other_ms = ctx->render_ms = ctx->host_wait_ms = ctx->walk_ms = ctx->packet_ms = 0.0;
    ctx->walk_launches = ctx->packet_launches = 0;
    ctx->untraced_paths = 0;
    ctx->bounce_extend_ms.clear();
    ctx->bounce_shade_ms.clear();
    ctx->profiled_batches = 0;
    return PT_OK;
}


#ifdef PT_WALK_STATS
// Debug build only: the lane-occupancy totals of the walking kernels since the last reset (tools/walk_stats.py).
extern "C" int ptc_debug_walk_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ptd::g_walk_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ptd::g_walk_stats), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
