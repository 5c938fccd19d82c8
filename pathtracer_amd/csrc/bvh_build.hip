// Linear BVH over the triangles of one buffer, built on the device: Morton order of the box
// centres (63 bits, ties broken by position), one radix sort, the hierarchy of Karras'
// "Maximizing parallelism in the construction of BVHs, octrees and k-d trees" (2012) with every
// internal node found independently, boxes fitted bottom-up, and the result written straight in
// the 64-byte two-box layout the traversal reads (dev_scene.h), with subtrees of at most LEAF_MAX
// triangles folded into leaves.  Triangle boxes are Triangle::bbox of the host layer
// (src/objects.rs bbox + BBox::pad, src/bvh.rs:51-65), so both builders enclose the same boxes.
#include "bvh_build.h"

#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "dev_scene.h"

namespace ptd {
namespace {

constexpr int BB = 256;
constexpr uint32_t LEAF_MAX = 4;
constexpr uint32_t IS_LEAF = 0x80000000u;  // in child links: the rest is a sorted triangle position

struct Box {
    float lo[3], hi[3];
};

// order-preserving map of floats onto unsigned integers, for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t u) {
    const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

// Triangle::bbox: componentwise min / max of the three positions, flat extents padded.
__device__ __forceinline__ Box tri_box(const float4 *cold, uint32_t i) {
    // the attribute matrix is stored row-major, row = component, column = vertex: x0 x1 x2 y0 y1 y2 z0 z1 z2 ...
    const float4 a = cold[6 * (size_t)i], b = cold[6 * (size_t)i + 1], c = cold[6 * (size_t)i + 2];
    const float x[3] = {a.x, a.y, a.z}, y[3] = {a.w, b.x, b.y}, z[3] = {b.z, b.w, c.x};
    Box r;
    r.lo[0] = fminf(x[0], fminf(x[1], x[2]));
    r.lo[1] = fminf(y[0], fminf(y[1], y[2]));
    r.lo[2] = fminf(z[0], fminf(z[1], z[2]));
    r.hi[0] = fmaxf(x[0], fmaxf(x[1], x[2]));
    r.hi[1] = fmaxf(y[0], fmaxf(y[1], y[2]));
    r.hi[2] = fmaxf(z[0], fmaxf(z[1], z[2]));
    for (int k = 0; k < 3; k++)
        if (fabsf(r.hi[k] - r.lo[k]) < 0.0001f) {
            r.lo[k] -= 0.0001f;
            r.hi[k] += 0.0001f;
        }
    return r;
}

__global__ __launch_bounds__(BB) void k_centre_bounds(const float4 *cold, uint32_t n, uint32_t *bounds /* lo xyz, hi xyz as f2ord */) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * BB + threadIdx.x; i < n; i += gridDim.x * BB) {
        const Box b = tri_box(cold, i);
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * (b.lo[k] + b.hi[k]);
            lo[k] = fminf(lo[k], c);
            hi[k] = fmaxf(hi[k], c);
        }
    }
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_down(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_down(hi[k], o, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0u)
        for (int k = 0; k < 3; k++) {
            if (lo[k] <= hi[k]) {  // skips waves without triangles and centres that are NaN
                atomicMin(&bounds[k], f2ord(lo[k]));
                atomicMax(&bounds[3 + k], f2ord(hi[k]));
            }
        }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit i of v to bit 3 i
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ __launch_bounds__(BB) void k_morton(const float4 *cold, uint32_t n, const uint32_t *bounds, uint64_t *keys, uint32_t *ids) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const Box b = tri_box(cold, i);
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(bounds[k]), hi = ord2f(bounds[3 + k]);
        const float c = 0.5f * (b.lo[k] + b.hi[k]);
        const float extent = hi - lo;
        float t = extent > 0.0f ? (c - lo) / extent : 0.0f;
        t = t * 2097152.0f;
        // NaN compares false and lands in cell 0
        q[k] = t >= 2097151.0f ? 2097151u : (t > 0.0f ? (uint32_t)t : 0u);
    }
    keys[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    ids[i] = i;
}

// boxes of the triangles in sorted order, 2 float4 each: (lo, hi.x) (hi.yz, -, -)
__global__ __launch_bounds__(BB) void k_leaf_boxes(const float4 *cold, uint32_t n, const uint32_t *order, float4 *leaf_box) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const Box b = tri_box(cold, order[i]);
    leaf_box[2 * (size_t)i] = make_float4(b.lo[0], b.lo[1], b.lo[2], b.hi[0]);
    leaf_box[2 * (size_t)i + 1] = make_float4(b.hi[1], b.hi[2], 0.0f, 0.0f);
}

// length of the common prefix of the keys at sorted positions i and j; equal keys are told apart by position
__device__ __forceinline__ int prefix(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = keys[i] ^ keys[j];
    if (x != 0ull) return __clzll((long long)x);
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

struct Links {
    uint32_t *child;   // [2 * (n - 1)] left, right of each internal node (IS_LEAF | position, or node index)
    uint32_t *parent;  // [2 n - 1] internal nodes first, then leaves; 0xFFFFFFFF for the root
    uint32_t *first;   // [n - 1] first sorted position below the node
    uint32_t *last;    // [n - 1] last one
};

__global__ __launch_bounds__(BB) void k_hierarchy(const uint64_t *keys, int n, Links L) {
    const int i = (int)(blockIdx.x * BB + threadIdx.x);
    if (i >= n - 1) return;
    const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t left = lo == gamma ? (IS_LEAF | (uint32_t)gamma) : (uint32_t)gamma;
    const uint32_t right = hi == gamma + 1 ? (IS_LEAF | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
    L.child[2 * i] = left;
    L.child[2 * i + 1] = right;
    L.first[i] = (uint32_t)lo;
    L.last[i] = (uint32_t)hi;
    L.parent[(left & IS_LEAF) ? (uint32_t)(n - 1) + (left & ~IS_LEAF) : left] = (uint32_t)i;
    L.parent[(right & IS_LEAF) ? (uint32_t)(n - 1) + (right & ~IS_LEAF) : right] = (uint32_t)i;
    if (i == 0) L.parent[0] = 0xFFFFFFFFu;
}

// Values another workgroup wrote earlier in this launch are read and written past the caches
// that are private to a compute unit or a die.
__device__ __forceinline__ float ld_shared(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_shared(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One thread per triangle walks towards the root; the second thread to arrive at a node fits its box.
// node_box: 8 floats per internal node (lo, hi, height as float, -).
__global__ __launch_bounds__(BB) void k_fit(uint32_t n, Links L, const float4 *leaf_box, float *node_box, uint32_t *arrived) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    uint32_t node = L.parent[(n - 1) + i];
    while (node != 0xFFFFFFFFu) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;  // the sibling subtree is not done yet
        __threadfence();
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, height = 0.0f;
        for (int c = 0; c < 2; c++) {
            const uint32_t ch = L.child[2 * node + c];
            float b[6], h;
            if (ch & IS_LEAF) {
                const float4 a = leaf_box[2 * (size_t)(ch & ~IS_LEAF)], e = leaf_box[2 * (size_t)(ch & ~IS_LEAF) + 1];
                b[0] = a.x, b[1] = a.y, b[2] = a.z, b[3] = a.w, b[4] = e.x, b[5] = e.y;
                h = 1.0f;
            } else {
                for (int k = 0; k < 6; k++) b[k] = ld_shared(&node_box[8 * (size_t)ch + k]);
                h = ld_shared(&node_box[8 * (size_t)ch + 6]);
            }
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], b[k]);
                hi[k] = fmaxf(hi[k], b[3 + k]);
            }
            height = fmaxf(height, h);
        }
        for (int k = 0; k < 3; k++) {
            st_shared(&node_box[8 * (size_t)node + k], lo[k]);
            st_shared(&node_box[8 * (size_t)node + 3 + k], hi[k]);
        }
        st_shared(&node_box[8 * (size_t)node + 6], height + 1.0f);
        node = L.parent[node];
    }
}

// Node i of the hierarchy becomes 64-byte node i.  Children that hold at most LEAF_MAX triangles
// become leaf references (the nodes below them are then never reached).  The child whose centre is
// lower along the axis on which the two centres differ most goes left, and that axis is recorded:
// the traversal visits the right child first when the ray points down that axis.
__global__ __launch_bounds__(BB) void k_emit(uint32_t n, Links L, const float4 *leaf_box, const float *node_box, float4 *wide) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    float box[2][6];
    uint32_t ref[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t ch = L.child[2 * i + c];
        if (ch & IS_LEAF) {
            const uint32_t p = ch & ~IS_LEAF;
            const float4 a = leaf_box[2 * (size_t)p], e = leaf_box[2 * (size_t)p + 1];
            box[c][0] = a.x, box[c][1] = a.y, box[c][2] = a.z, box[c][3] = a.w, box[c][4] = e.x, box[c][5] = e.y;
            ref[c] = WIDE_LEAF | (1u << 24) | p;
        } else {
            for (int k = 0; k < 6; k++) box[c][k] = node_box[8 * (size_t)ch + k];
            const uint32_t count = L.last[ch] - L.first[ch] + 1u;
            ref[c] = count <= LEAF_MAX ? (WIDE_LEAF | (count << 24) | L.first[ch]) : ch;
        }
    }
    uint32_t axis = 0;
    float best = -1.0f;
    for (uint32_t k = 0; k < 3; k++) {
        const float sep = fabsf((box[0][k] + box[0][3 + k]) - (box[1][k] + box[1][3 + k]));
        if (sep > best) {
            best = sep;
            axis = k;
        }
    }
    const int l = (box[0][axis] + box[0][3 + axis]) <= (box[1][axis] + box[1][3 + axis]) ? 0 : 1, r = l ^ 1;
    float4 *w = wide + 4 * (size_t)i;
    // the layout of a 64-byte node's two boxes (dev_traverse.h): x and y of a corner side by side, the z of both corners side by side
    w[0] = make_float4(box[l][0], box[l][1], box[l][3], box[l][4]);
    w[1] = make_float4(box[l][2], box[l][5], box[r][2], box[r][5]);
    w[2] = make_float4(box[r][0], box[r][1], box[r][3], box[r][4]);
    w[3] = make_float4(__uint_as_float(ref[l]), __uint_as_float(ref[r]), __uint_as_float(axis), 0.0f);
}

template <class T, int PER>
__global__ __launch_bounds__(BB) void k_gather(const T *src, T *dst, const uint32_t *order, uint32_t n) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < PER; k++) dst[(size_t)PER * i + k] = src[(size_t)PER * order[i] + k];
}

struct Temp {
    std::vector<void *> ptrs;
    ~Temp() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T>
    hipError_t get(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) {
            ptrs.push_back(p);
            *out = (T *)p;
        }
        return e;
    }
};

}  // namespace

#define BUILD_TRY(expr)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                   \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int build_tri_bvh_device(hipStream_t stream, float4 *tri_hot, float4 *tri_cold, int32_t *tri_mat, uint32_t n, float4 *wide, uint32_t *order,
                         TriBvhResult *out, std::string &err) {
    if (n == 0 || n > (1u << 24)) {
        err = "device BVH build takes 1 .. 2^24 triangles per buffer";
        return 1;
    }
    Temp tmp;
    const int grid_n = (int)((n + BB - 1) / BB);
    uint32_t *bounds;
    BUILD_TRY(tmp.get(&bounds, 6));
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    BUILD_TRY(hipMemcpyAsync(bounds, init, sizeof init, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_centre_bounds, dim3(std::min(grid_n, 2048)), dim3(BB), 0, stream, tri_cold, n, bounds);
    uint64_t *keys_in, *keys;
    uint32_t *ids_in, *ids;
    BUILD_TRY(tmp.get(&keys_in, n));
    BUILD_TRY(tmp.get(&keys, n));
    BUILD_TRY(tmp.get(&ids_in, n));
    BUILD_TRY(tmp.get(&ids, n));
    hipLaunchKernelGGL(k_morton, dim3(grid_n), dim3(BB), 0, stream, tri_cold, n, bounds, keys_in, ids_in);
    size_t sort_bytes = 0;
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_in, keys, ids_in, ids, (size_t)n, 0u, 63u, stream));
    char *sort_tmp;
    BUILD_TRY(tmp.get(&sort_tmp, sort_bytes));
    BUILD_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys_in, keys, ids_in, ids, (size_t)n, 0u, 63u, stream));
    float4 *leaf_box;
    BUILD_TRY(tmp.get(&leaf_box, 2 * (size_t)n));
    hipLaunchKernelGGL(k_leaf_boxes, dim3(grid_n), dim3(BB), 0, stream, tri_cold, n, ids, leaf_box);

    TriBvhResult res{};
    std::vector<float4> host_boxes;
    if (n <= LEAF_MAX) {
        // a handful of triangles: one leaf, no nodes
        host_boxes.resize(2 * (size_t)n);
        BUILD_TRY(hipMemcpyAsync(host_boxes.data(), leaf_box, host_boxes.size() * sizeof(float4), hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        for (int k = 0; k < 3; k++) res.rmin[k] = INFINITY, res.rmax[k] = -INFINITY;
        for (uint32_t i = 0; i < n; i++) {
            const float4 a = host_boxes[2 * i], e = host_boxes[2 * i + 1];
            const float lo[3] = {a.x, a.y, a.z}, hi[3] = {a.w, e.x, e.y};
            for (int k = 0; k < 3; k++) {
                res.rmin[k] = std::fmin(res.rmin[k], lo[k]);
                res.rmax[k] = std::fmax(res.rmax[k], hi[k]);
            }
        }
        res.root_ref = WIDE_LEAF | (n << 24);
        res.n_nodes = 0;
        res.height = 1;
    } else {
        Links L{};
        BUILD_TRY(tmp.get(&L.child, 2 * (size_t)(n - 1)));
        BUILD_TRY(tmp.get(&L.parent, 2 * (size_t)n - 1));
        BUILD_TRY(tmp.get(&L.first, (size_t)(n - 1)));
        BUILD_TRY(tmp.get(&L.last, (size_t)(n - 1)));
        float *node_box;
        uint32_t *arrived;
        BUILD_TRY(tmp.get(&node_box, 8 * (size_t)(n - 1)));
        BUILD_TRY(tmp.get(&arrived, (size_t)(n - 1)));
        BUILD_TRY(hipMemsetAsync(arrived, 0, (size_t)(n - 1) * sizeof(uint32_t), stream));
        const int grid_i = (int)((n - 1 + BB - 1) / BB);
        hipLaunchKernelGGL(k_hierarchy, dim3(grid_i), dim3(BB), 0, stream, keys, (int)n, L);
        hipLaunchKernelGGL(k_fit, dim3(grid_n), dim3(BB), 0, stream, n, L, leaf_box, node_box, arrived);
        hipLaunchKernelGGL(k_emit, dim3(grid_i), dim3(BB), 0, stream, n, L, leaf_box, node_box, wide);
        float root[8];
        BUILD_TRY(hipMemcpyAsync(root, node_box, sizeof root, hipMemcpyDeviceToHost, stream));
        BUILD_TRY(hipStreamSynchronize(stream));
        for (int k = 0; k < 3; k++) res.rmin[k] = root[k], res.rmax[k] = root[3 + k];
        res.root_ref = 0;
        res.n_nodes = n - 1;
        res.height = (uint32_t)root[6];
    }
    // the triangles themselves, in sorted order
    {
        float4 *hot2, *cold2;
        int32_t *mat2;
        BUILD_TRY(tmp.get(&hot2, 3 * (size_t)n));
        BUILD_TRY(tmp.get(&cold2, 6 * (size_t)n));
        BUILD_TRY(tmp.get(&mat2, (size_t)n));
        hipLaunchKernelGGL((k_gather<float4, 3>), dim3(grid_n), dim3(BB), 0, stream, tri_hot, hot2, ids, n);
        hipLaunchKernelGGL((k_gather<float4, 6>), dim3(grid_n), dim3(BB), 0, stream, tri_cold, cold2, ids, n);
        hipLaunchKernelGGL((k_gather<int32_t, 1>), dim3(grid_n), dim3(BB), 0, stream, tri_mat, mat2, ids, n);
        BUILD_TRY(hipMemcpyAsync(tri_hot, hot2, 3 * (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        BUILD_TRY(hipMemcpyAsync(tri_cold, cold2, 6 * (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        BUILD_TRY(hipMemcpyAsync(tri_mat, mat2, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
        if (order) {  // the caller's names of the triangles move with them
            uint32_t *order2;
            BUILD_TRY(tmp.get(&order2, (size_t)n));
            hipLaunchKernelGGL((k_gather<uint32_t, 1>), dim3(grid_n), dim3(BB), 0, stream, order, order2, ids, n);
            BUILD_TRY(hipMemcpyAsync(order, order2, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        }
        BUILD_TRY(hipStreamSynchronize(stream));
    }
    BUILD_TRY(hipGetLastError());
    *out = res;
    return 0;
}

}  // namespace ptd
