// place_exp10.hip — does the ALLOCATION TYPE change the write-bandwidth class? hipExtMallocWithFlags with
// default / fine-grained / uncached / contiguous, NA allocations of 3.44 GB each held side by side, 2 warm sweeps and a
// timed sweep each; pure streaming stores (wr) and the von Mises read:write mix (mix: 13 rows read from one input slab,
// 43 rows written per tile). build: hipcc --offload-arch=gfx950 -O3 -o place_exp10 place_exp10.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

__global__ __launch_bounds__(256) void wr(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        f64x2* d = dst + t * (43 * 64);
#pragma unroll
        for (int k = 0; k < 43; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
    }
}
__global__ __launch_bounds__(256) void mix(long n_tiles, const f64x2* __restrict__ src, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        const f64x2* s = src + t * (13 * 64);
        f64x2 acc = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 13; ++k) acc += s[k * 64 + lane];
        f64x2* d = dst + t * (43 * 64);
#pragma unroll
        for (int k = 0; k < 43; ++k) __builtin_nontemporal_store(acc + (double)k, d + k * 64 + lane);
    }
}
static hipStream_t st;
float timeit(int launches, int reps, const std::function<void()>& fn) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> v;
    fn(); fn();
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, st)); for (int l = 0; l < launches; ++l) fn(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms / launches);
    }
    std::sort(v.begin(), v.end()); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[v.size() / 2];
}
int main(int argc, char** argv) {
    const int NA = argc > 1 ? atoi(argv[1]) : 6;
    const long tiles = 10000000 / 64;                       // 64-point tiles of the 10^7-point von Mises batch
    const size_t out_bytes = (size_t)tiles * 43 * 1024, in_bytes = (size_t)tiles * 13 * 1024;
    CK(hipStreamCreate(&st));
    void* src = nullptr;
    CK(hipMalloc(&src, in_bytes));
    CK(hipMemset(src, 0, in_bytes));
    const struct { const char* name; unsigned flag; } kinds[] = {{"default", hipDeviceMallocDefault}, {"contiguous", hipDeviceMallocContiguous},
                                                                {"uncached", hipDeviceMallocUncached}, {"finegrained", hipDeviceMallocFinegrained}};
    for (int round = 0; round < 2; ++round)
        for (const auto& k : kinds) {
            std::vector<void*> a(NA, nullptr);
            int got = 0;
            for (auto& p : a) {
                if (hipExtMallocWithFlags(&p, out_bytes, k.flag) != hipSuccess) { (void)hipGetLastError(); p = nullptr; break; }
                ++got;
            }
            printf("%-11s (%d allocations) write GB/s:", k.name, got);
            for (int i = 0; i < got; ++i) {
                float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, tiles, (f64x2*)a[i]); });
                printf(" %5.0f", out_bytes / ms / 1e6);
            }
            printf("\n%-11s                 mix   GB/s:", "");
            for (int i = 0; i < got; ++i) {
                float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(4096), dim3(256), 0, st, tiles, (const f64x2*)src, (f64x2*)a[i]); });
                printf(" %5.0f", (out_bytes + in_bytes) / ms / 1e6);
            }
            printf("\n");
            fflush(stdout);
            for (auto p : a) if (p) CK(hipFree(p));
        }
    return 0;
}
