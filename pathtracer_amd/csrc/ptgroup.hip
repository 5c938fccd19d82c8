// pt_group: Camera::render (src/camera.rs:162-236) spread over several GPUs of one node, behind one
// handle.  The passes of the reference's loop (src/camera.rs:185-198) are independent once the
// generator is keyed by (seed, pixel, stratum); all they share is the additive accumulator
// (:172-173, :216) and the pass count (:199).  So every member context holds the whole scene, a
// render call cuts its stratum range into contiguous pieces, one per member, each member
// enqueues its piece from a host thread of its own, and a reader sums the members' accumulators
// onto the first member's GPU with ONE reduce (RCCL: ncclReduce, sum, f32, W*H*3 values, over
// xGMI).  No other collective exists; the pass counts are known on the host.
//
// RCCL is opened at run time (dlopen), so that single-GPU users of libptcore do not load it and a
// process that also carries another copy of the library (PyTorch ships one) does not mix the two.
// A group that lists one device more than once cannot form an RCCL communicator; its members are
// then summed in member order with copies and an add kernel, which is also how the tests exercise
// the split and the merge on a machine with a single GPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "ptcore_internal.h"

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool open(std::string &why) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("librccl.so not found: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Reduce || !GetErrorString) {
            why = "librccl.so lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclReduce / ncclGetErrorString";
            return false;
        }
        return true;
    }
};

__global__ void k_add_into(float *dst, const float *src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

}  // namespace

struct pt_group {
    std::vector<pt_context *> members;
    std::vector<int> devices;
    bool distinct = true;  // no device listed twice
    // The merge is an RCCL reduce when the group has several distinct devices.  Opening librccl and making the
    // communicators takes seconds, so it runs on a thread of its own from pt_group_create on and is waited for by the
    // first merge: scene upload and rendering do not wait for it.  A group of one device needs no collective (its
    // merge is a copy), unless PT_GROUP_RCCL_ALWAYS is set in the environment, which keeps the one-rank reduce
    // available for testing the call path on a machine with a single GPU.
    bool use_rccl = false;
    std::thread rccl_thread;
    int rccl_rc = PT_OK;
    std::string rccl_err;
    Rccl rccl;
    std::vector<ncclComm_t> comms;
    float *reduced = nullptr;  // on the first member's device: the sum of all accumulators, W*H*3 f32
    float *staging = nullptr;  // likewise, for the merge without RCCL
    uint64_t film_pixels = 0;  // pixels `reduced` / `staging` are allocated for
    bool film_pixels_valid = false;  // every member has the camera those buffers belong to
    std::string err;
};

static thread_local std::string g_group_create_error;

static int gfail(pt_group *g, int code, const char *fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    else g_group_create_error = buf;
    return code;
}
// A member's own error, with its place in the group.
static int member_fail(pt_group *g, size_t i, int rc) {
    return gfail(g, rc, "member %zu (device %d): %s", i, g->devices[i], pt_last_error(g->members[i]));
}

void pt_group_split(uint32_t stratum_begin, uint32_t stratum_end, int n, int i, uint32_t *begin_i, uint32_t *end_i) {
    // contiguous pieces whose sizes differ by at most one, the larger ones first
    const uint32_t total = stratum_end > stratum_begin ? stratum_end - stratum_begin : 0u;
    if (n <= 0 || i < 0 || i >= n) {
        if (begin_i) *begin_i = stratum_begin;
        if (end_i) *end_i = stratum_begin;
        return;
    }
    const uint32_t q = total / (uint32_t)n, r = total % (uint32_t)n, k = (uint32_t)i;
    const uint32_t b = stratum_begin + k * q + std::min(k, r);
    if (begin_i) *begin_i = b;
    if (end_i) *end_i = b + q + (k < r ? 1u : 0u);
}

// Waits for the RCCL set-up started by pt_group_create; its result.
static int rccl_wait(pt_group *g) {
    if (g->rccl_thread.joinable()) g->rccl_thread.join();
    return g->rccl_rc;
}

void pt_group_destroy(pt_group *g) {
    if (!g) return;
    (void)rccl_wait(g);
    for (size_t i = 0; i < g->comms.size(); i++)
        if (g->comms[i]) (void)g->rccl.CommDestroy(g->comms[i]);
    if (!g->members.empty()) {
        (void)hipSetDevice(g->devices[0]);
        if (g->reduced) (void)hipFree(g->reduced);
        if (g->staging) (void)hipFree(g->staging);
    }
    for (pt_context *c : g->members) pt_destroy(c);
    if (g->rccl.lib) dlclose(g->rccl.lib);
    delete g;
}

int pt_group_create(const int *devices, int n, pt_group **out) {
    if (!out) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n <= 0 || n > 64) return gfail(nullptr, PT_ERR_INVALID, "pt_group_create: 1 to 64 devices");
    pt_group *g = new (std::nothrow) pt_group();
    if (!g) return gfail(nullptr, PT_ERR_NOMEM, "out of host memory");
    g->devices.assign(devices, devices + n);
    g->distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == (size_t)n;
    // one host thread per member: creating a context (streams, events, pinned words) takes a while, and the
    // members do not depend on each other
    g->members.assign((size_t)n, nullptr);
    std::vector<int> rcs((size_t)n, PT_OK);
    std::vector<std::string> errs((size_t)n);
    auto make = [&](int i) {
        rcs[(size_t)i] = pt_create(devices[i], &g->members[(size_t)i]);
        if (rcs[(size_t)i]) errs[(size_t)i] = pt_last_error(nullptr);  // thread-local: read on the thread that failed
    };
    {
        std::vector<std::thread> pool;
        for (int i = 1; i < n; i++) pool.emplace_back(make, i);
        make(0);
        for (auto &t : pool) t.join();
    }
    for (int i = 0; i < n; i++)
        if (rcs[(size_t)i]) {
            const int rc = rcs[(size_t)i];
            gfail(nullptr, rc, "member %d (device %d): %s", i, devices[i], errs[(size_t)i].c_str());
            g->members.erase(std::remove(g->members.begin(), g->members.end(), nullptr), g->members.end());
            pt_group_destroy(g);
            return rc;
        }
    g->use_rccl = g->distinct && (n > 1 || getenv("PT_GROUP_RCCL_ALWAYS") != nullptr);
    if (g->use_rccl) {
        g->comms.assign((size_t)n, nullptr);
        g->rccl_thread = std::thread([g, n] {
            std::string why;
            if (!g->rccl.open(why)) {
                g->rccl_err = "pt_group_create: " + why;
                g->rccl_rc = PT_ERR_UNSUPPORTED;
                g->comms.clear();
                return;
            }
            const ncclResult_t e = g->rccl.CommInitAll(g->comms.data(), n, g->devices.data());
            if (e != ncclSuccess) {
                g->rccl_err = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(e);
                g->rccl_rc = PT_ERR_HIP;
                g->comms.clear();
            }
        });
    }
    *out = g;
    return PT_OK;
}

const char *pt_group_last_error(const pt_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
int pt_group_size(const pt_group *g) { return g ? (int)g->members.size() : 0; }
pt_context *pt_group_member(pt_group *g, int i) { return g && i >= 0 && (size_t)i < g->members.size() ? g->members[(size_t)i] : nullptr; }

// fn(member) on every member, each on a host thread of its own; the first failure (in member order) is reported
// after all have finished, so that no member is left half-way while another is still working.
template <class Fn>
static int for_each_member(pt_group *g, Fn &&fn) {
    const size_t n = g->members.size();
    std::vector<int> rcs(n, PT_OK);
    std::vector<std::thread> pool;
    for (size_t i = 1; i < n; i++) pool.emplace_back([&, i] { rcs[i] = fn(g->members[i]); });
    rcs[0] = fn(g->members[0]);
    for (auto &t : pool) t.join();
    for (size_t i = 0; i < n; i++)
        if (rcs[i]) return member_fail(g, i, rcs[i]);
    return PT_OK;
}

int pt_group_set_option(pt_group *g, int key, double value) {
    if (!g) return PT_ERR_INVALID;
    // an option value is either valid for every member or for none (they are the same library on the same kind of
    // device), so a failure leaves no member changed
    return for_each_member(g, [&](pt_context *c) { return pt_set_option(c, key, value); });
}

int pt_group_upload_scene(pt_group *g, const pt_scene_desc *scene) {
    if (!g) return PT_ERR_INVALID;
    // every GPU holds the whole scene (the largest benchmark scene is under 200 MB)
    return for_each_member(g, [&](pt_context *c) { return pt_upload_scene(c, scene); });
}

int pt_group_set_camera(pt_group *g, const pt_camera_desc *camera) {
    if (!g) return PT_ERR_INVALID;
    if (int rc = for_each_member(g, [&](pt_context *c) { return pt_set_camera(c, camera); })) {
        g->film_pixels_valid = false;  // some members may have the new camera and some not: the group has none
        return rc;
    }
    g->film_pixels_valid = true;
    const uint64_t npix = ptc_npix(g->members[0]);
    if (npix != g->film_pixels) {
        if (hipSetDevice(g->devices[0]) != hipSuccess) return gfail(g, PT_ERR_HIP, "hipSetDevice(%d) failed", g->devices[0]);
        if (g->reduced) (void)hipFree(g->reduced);
        if (g->staging) (void)hipFree(g->staging);
        g->reduced = g->staging = nullptr;
        g->film_pixels = 0;
        hipError_t e = hipMalloc((void **)&g->reduced, npix * 3 * sizeof(float));
        if (e == hipSuccess && !g->use_rccl && g->members.size() > 1) e = hipMalloc((void **)&g->staging, npix * 3 * sizeof(float));
        if (e != hipSuccess) {
            if (g->reduced) (void)hipFree(g->reduced);
            g->reduced = nullptr;
            return gfail(g, PT_ERR_NOMEM, "hipMalloc of the merged accumulator: %s", hipGetErrorString(e));
        }
        g->film_pixels = npix;
    }
    return PT_OK;
}

int pt_group_render(pt_group *g, uint32_t stratum_begin, uint32_t stratum_end, uint64_t seed) {
    if (!g) return PT_ERR_INVALID;
    if (stratum_begin > stratum_end) return gfail(g, PT_ERR_INVALID, "stratum range [%u,%u) is reversed", stratum_begin, stratum_end);
    const int n = (int)g->members.size();
    std::vector<int> rcs((size_t)n, PT_OK);
    auto piece = [&](int i) {
        uint32_t b, e;
        pt_group_split(stratum_begin, stratum_end, n, i, &b, &e);
        rcs[(size_t)i] = b < e ? pt_render(g->members[(size_t)i], b, e, seed) : PT_OK;
    };
    // pt_render enqueues and returns, but waits when it is several launches ahead of its GPU: a thread per
    // member keeps one member's wait from holding back the others
    std::vector<std::thread> pool;
    for (int i = 1; i < n; i++) pool.emplace_back(piece, i);
    piece(0);
    for (auto &t : pool) t.join();
    for (int i = 0; i < n; i++)
        if (rcs[(size_t)i]) return member_fail(g, (size_t)i, rcs[(size_t)i]);
    return PT_OK;
}

int pt_group_sync(pt_group *g) {
    if (!g) return PT_ERR_INVALID;
    return for_each_member(g, [](pt_context *c) { return pt_sync(c); });
}

// Sum of the members' accumulators into g->reduced on the first member's device; *n_samples = passes in it.
static int merge(pt_group *g, uint32_t *n_samples) {
    if (!g->film_pixels || !g->film_pixels_valid) return gfail(g, PT_ERR_STATE, "no camera set");
    if (int rc = pt_group_sync(g)) return rc;
    const size_t count = (size_t)g->film_pixels * 3;
    uint32_t total = 0;
    for (pt_context *c : g->members) total += ptc_n_samples(c);
    *n_samples = total;
    const int n = (int)g->members.size();
#define G_HIP(expr)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return gfail(g, PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
    if (g->use_rccl) {
        if (int rc = rccl_wait(g)) return gfail(g, rc, "%s", g->rccl_err.c_str());
        // whatever fails between GroupStart and GroupEnd, GroupEnd is still called: an open group would nest every
        // later RCCL call of the process
        ncclResult_t e = g->rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (int i = 0; i < n && e == ncclSuccess && he == hipSuccess; i++) {
            he = hipSetDevice(g->devices[(size_t)i]);
            if (he != hipSuccess) break;
            pt_context *c = g->members[(size_t)i];
            e = g->rccl.Reduce(ptc_accum(c), i == 0 ? g->reduced : nullptr, count, ncclFloat, ncclSum, 0, g->comms[(size_t)i], ptc_stream(c));
        }
        const ncclResult_t e2 = g->rccl.GroupEnd();
        if (he != hipSuccess) return gfail(g, PT_ERR_HIP, "hipSetDevice before ncclReduce failed: %s", hipGetErrorString(he));
        if (e == ncclSuccess) e = e2;
        if (e != ncclSuccess) return gfail(g, PT_ERR_HIP, "ncclReduce of the accumulators: %s", g->rccl.GetErrorString(e));
        for (int i = 0; i < n; i++) {
            G_HIP(hipSetDevice(g->devices[(size_t)i]));
            G_HIP(hipStreamSynchronize(ptc_stream(g->members[(size_t)i])));
        }
    } else {
        G_HIP(hipSetDevice(g->devices[0]));
        hipStream_t st = ptc_stream(g->members[0]);
        G_HIP(hipMemcpyAsync(g->reduced, ptc_accum(g->members[0]), count * sizeof(float), hipMemcpyDeviceToDevice, st));
        for (int i = 1; i < n; i++) {
            G_HIP(hipMemcpyPeerAsync(g->staging, g->devices[0], ptc_accum(g->members[(size_t)i]), g->devices[(size_t)i], count * sizeof(float), st));
            hipLaunchKernelGGL(k_add_into, dim3(2048), dim3(256), 0, st, g->reduced, g->staging, count);
        }
        G_HIP(hipStreamSynchronize(st));
        G_HIP(hipGetLastError());
    }
#undef G_HIP
    return PT_OK;
}

int pt_group_read_accum(pt_group *g, float *rgb_sum, uint32_t *n_samples) {
    if (!g) return PT_ERR_INVALID;
    uint32_t total = 0;
    if (int rc = merge(g, &total)) return rc;
    if (rgb_sum) {
        if (hipSetDevice(g->devices[0]) != hipSuccess) return gfail(g, PT_ERR_HIP, "hipSetDevice(%d) failed", g->devices[0]);
        const hipError_t e = hipMemcpy(rgb_sum, g->reduced, (size_t)g->film_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return gfail(g, PT_ERR_HIP, "copy of the merged accumulator: %s", hipGetErrorString(e));
    }
    if (n_samples) *n_samples = total;
    return PT_OK;
}

int pt_group_read_rgb8(pt_group *g, uint8_t *rgb) {
    if (!g || !rgb) return gfail(g, PT_ERR_INVALID, "NULL argument");
    uint32_t total = 0;
    if (int rc = merge(g, &total)) return rc;
    if (int rc = ptc_tonemap(g->members[0], g->reduced, total, rgb)) return member_fail(g, 0, rc);
    return PT_OK;
}

int pt_group_reset_accum(pt_group *g) {
    if (!g) return PT_ERR_INVALID;
    for (size_t i = 0; i < g->members.size(); i++)
        if (int rc = pt_reset_accum(g->members[i])) return member_fail(g, i, rc);
    return PT_OK;
}

int pt_group_stats_get(pt_group *g, pt_stats *out) {
    if (!g || !out) return gfail(g, PT_ERR_INVALID, "NULL argument");
    std::memset(out, 0, sizeof *out);
    for (size_t i = 0; i < g->members.size(); i++) {
        pt_stats s;
        if (int rc = pt_stats_get(g->members[i], &s)) return member_fail(g, i, rc);
        out->rays += s.rays;
        out->paths += s.paths;
        out->extend_launches += s.extend_launches;
        out->node_tests += s.node_tests;
        out->tri_tests += s.tri_tests;
        out->sphere_tests += s.sphere_tests;
        out->quad_tests += s.quad_tests;
        out->device_bytes += s.device_bytes;
        out->bad_refs += s.bad_refs;
        // the members work side by side: times are those of the slowest
        out->extend_ms = std::max(out->extend_ms, s.extend_ms);
        out->extend_busy_ms = std::max(out->extend_busy_ms, s.extend_busy_ms);
        out->shade_ms = std::max(out->shade_ms, s.shade_ms);
        out->other_ms = std::max(out->other_ms, s.other_ms);
        out->render_ms = std::max(out->render_ms, s.render_ms);
        out->host_wait_ms = std::max(out->host_wait_ms, s.host_wait_ms);
        out->walk_ms = std::max(out->walk_ms, s.walk_ms);
        out->packet_ms = std::max(out->packet_ms, s.packet_ms);
        out->walk_launches += s.walk_launches;
        out->packet_launches += s.packet_launches;
        out->shadow_rays += s.shadow_rays;
    }
    return PT_OK;
}

int pt_group_stats_reset(pt_group *g) {
    if (!g) return PT_ERR_INVALID;
    for (size_t i = 0; i < g->members.size(); i++)
        if (int rc = pt_stats_reset(g->members[i])) return member_fail(g, i, rc);
    return PT_OK;
}
