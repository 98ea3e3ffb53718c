// place_pmc.hip — counters for a FAST and a SLOW virtual range holding the same kind of memory (place_exp7: the
// speed of a 3 GB streaming-write sweep follows the virtual address). 12 slabs of 3 GB (hipMemCreate chunks mapped
// side by side); the fastest and slowest are then swept by two kernels with distinct names, so that
// `rocprofv3 --pmc ...` reports their counters separately (scripts/profile_placement.sh).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

__device__ __forceinline__ void sweep(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        f64x2* d = dst + t * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
    }
}
__global__ __launch_bounds__(256) void wr_probe(long n, f64x2* dst) { sweep(n, dst); }
__global__ __launch_bounds__(256) void wr_fast_range(long n, f64x2* dst) { sweep(n, dst); }
__global__ __launch_bounds__(256) void wr_slow_range(long n, f64x2* dst) { sweep(n, dst); }

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, GB = 1ull << 30, CH = 128 * MB, B3 = 3 * GB;
    const int NS = 12;
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char* R = nullptr;
    CK(hipMemAddressReserve((void**)&R, (size_t)NS * B3, 2 * MB, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(NS * 24);
    for (size_t i = 0; i < h.size(); ++i) { CK(hipMemCreate(&h[i], CH, &prop, 0)); CK(hipMemMap(R + CH * i, CH, 0, h[i], 0)); }
    CK(hipMemSetAccess(R, (size_t)NS * B3, &acc, 1));
    const long n = B3 / 16384;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<double> w(NS);
    for (int s = 0; s < NS; ++s) {
        f64x2* p = (f64x2*)(R + (size_t)s * B3);
        hipLaunchKernelGGL(wr_probe, dim3(4096), dim3(256), 0, st, n, p);
        CK(hipEventRecord(a, st));
        for (int l = 0; l < 4; ++l) hipLaunchKernelGGL(wr_probe, dim3(4096), dim3(256), 0, st, n, p);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        w[s] = B3 / (ms / 4) / 1e6;
        printf("slab %2d: %.0f GB/s\n", s, w[s]);
    }
    const int f = (int)(std::max_element(w.begin(), w.end()) - w.begin()), s = (int)(std::min_element(w.begin(), w.end()) - w.begin());
    printf("fast range = slab %d (%.0f), slow range = slab %d (%.0f)\n", f, w[f], s, w[s]);
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(wr_fast_range, dim3(4096), dim3(256), 0, st, n, (f64x2*)(R + (size_t)f * B3));
        hipLaunchKernelGGL(wr_slow_range, dim3(4096), dim3(256), 0, st, n, (f64x2*)(R + (size_t)s * B3));
    }
    CK(hipStreamSynchronize(st));
    return 0;
}
