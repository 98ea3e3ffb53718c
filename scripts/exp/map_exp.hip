// map_exp.hip — write/read bandwidth of consecutive 3 GB hipMalloc allocations (placement map).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ __launch_bounds__(256) void wr(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        f64x2* d = dst + t * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
    }
}
__global__ __launch_bounds__(256) void rd(long n_tiles, const f64x2* __restrict__ src, f64x2* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f64x2 acc = {0, 0};
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        const f64x2* s = src + t * (16 * 64);
        f64x2 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = s[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc.x == 1.234e300) sink[lane] = acc;
}
float timeit(hipStream_t st, const std::function<void()>& fn) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> v;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(a, st)); for (int l = 0; l < 4; ++l) fn(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms / 4);
    }
    std::sort(v.begin(), v.end()); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[1];
}
int main(int argc, char** argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 40;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t bytes = 3000L * 1000 * 1000 / 16384 * 16384;
    const long n_tiles = bytes / 16384;
    f64x2* sink; CK(hipMalloc(&sink, 4096));
    std::vector<char*> bufs(nbuf);
    const bool one_big = argc > 2;   // second argument: carve all buffers out of ONE allocation
    if (one_big) {
        char* big; CK(hipMalloc(&big, bytes * (size_t)nbuf + (1ull << 30)));
        printf("one allocation of %.1f GB at %p\n", (bytes * (double)nbuf) / 1e9, (void*)big);
        for (int i = 0; i < nbuf; ++i) bufs[i] = big + bytes * (size_t)i;
    } else {
        for (auto& b : bufs) CK(hipMalloc(&b, bytes));
    }
    for (int i = 0; i < nbuf; ++i) {
        f64x2* p = (f64x2*)bufs[i];
        float w = timeit(st, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, n_tiles, p); });
        float r = timeit(st, [&] { hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, st, n_tiles, (const f64x2*)p, sink); });
        printf("buf %2d va %p write %7.1f read %7.1f GB/s\n", i, (void*)p, bytes / w / 1e6, bytes / r / 1e6);
    }
    return 0;
}
