// Throughput of wave-leader atomicAdd-with-return on a few hot addresses (what queue tails
// and cursors do), as a function of how many distinct addresses the waves spread over.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned *ctr, int n_addr, int stride_words, int iters, unsigned *sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const unsigned lane = threadIdx.x % 64;
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned v = 0;
        if (lane == 0) v = atomicAdd(&ctr[((wave + i) % n_addr) * stride_words], 64u);
        v = __shfl(v, 0, 64);
        acc += v;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
int main() {
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 1 << 20);
    hipMalloc(&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = 1024, iters = 64;
    const double total = (double)grid * 4 * iters;
    for (int stride : {1, 16, 64})
        for (int n_addr : {1, 2, 8, 64, 1024}) {
            hipMemset(ctr, 0, 1 << 20);
            k<<<grid, 256>>>(ctr, n_addr, stride, 2, sink);
            hipEventRecord(a);
            k<<<grid, 256>>>(ctr, n_addr, stride, iters, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("stride %3d words, %5d addresses: %8.3f ms for %.0f atomics = %7.2f ns each (%.1f M/s)\n", stride, n_addr, ms, total, ms * 1e6 / total, total / ms / 1e3);
        }
    return 0;
}
