// order_exp.hip — does the ORDER in which a streaming-write kernel sweeps its buffer matter?
// Pure NT write of 3 GB per buffer, 16-byte lane-linear, tile = 16 KiB per wave-iteration.
//   mode 0: linear (block b handles tiles b, b+G, ...), mode 1: bit-permuted tile index (multiplicative hash)
//   mode 2: tiles dealt round-robin over K large regions (region = tile % K, offset = tile / K)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void wr(long n_tiles, long mult, int K, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    for (long t = gw; t < n_tiles; t += nw) {
        long tile = t;
        if (MODE == 1) tile = (t * mult) % n_tiles;
        if (MODE == 2) { const long per = n_tiles / K; tile = (t % K) * per + (t / K); if (t >= per * K) tile = t; }
        f64x2* d = dst + tile * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
    }
}
float timeit(hipStream_t st, const std::function<void()>& fn) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> v;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, st)); for (int l = 0; l < 5; ++l) fn(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms / 5);
    }
    std::sort(v.begin(), v.end()); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[2];
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t bytes = 3000L * 1000 * 1000 / 16384 * 16384;
    const long n_tiles = bytes / 16384;
    const int nbuf = 6;
    std::vector<char*> bufs(nbuf);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0, bytes)); }
    const int full = (int)((n_tiles + 3) / 4);
    for (int i = 0; i < nbuf; ++i) {
        f64x2* p = (f64x2*)bufs[i];
        float a = timeit(st, [&] { hipLaunchKernelGGL(wr<0>, dim3(full), dim3(256), 0, st, n_tiles, 1L, 1, p); });
        float a2 = timeit(st, [&] { hipLaunchKernelGGL(wr<0>, dim3(4096), dim3(256), 0, st, n_tiles, 1L, 1, p); });
        float b = timeit(st, [&] { hipLaunchKernelGGL(wr<1>, dim3(full), dim3(256), 0, st, n_tiles, 7919L, 1, p); });
        float b2 = timeit(st, [&] { hipLaunchKernelGGL(wr<1>, dim3(full), dim3(256), 0, st, n_tiles, 104729L, 1, p); });
        float c = timeit(st, [&] { hipLaunchKernelGGL(wr<2>, dim3(full), dim3(256), 0, st, n_tiles, 1L, 8, p); });
        float c2 = timeit(st, [&] { hipLaunchKernelGGL(wr<2>, dim3(full), dim3(256), 0, st, n_tiles, 1L, 64, p); });
        float c3 = timeit(st, [&] { hipLaunchKernelGGL(wr<2>, dim3(full), dim3(256), 0, st, n_tiles, 1L, 1024, p); });
        printf("buf %d: linear %7.1f | linear bpc16 %7.1f | hash7919 %7.1f | hash104729 %7.1f | deal8 %7.1f | deal64 %7.1f | deal1024 %7.1f GB/s\n", i,
               bytes / a / 1e6, bytes / a2 / 1e6, bytes / b / 1e6, bytes / b2 / 1e6, bytes / c / 1e6, bytes / c2 / 1e6, bytes / c3 / 1e6);
    }
    return 0;
}
