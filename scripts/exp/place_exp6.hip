// place_exp6.hip — place_exp3..5: neither the physical chunks nor the virtual address alone decide whether a sweep is
// fast; window maps change sharply with the sweep's START. Model to test: channel/bank "camping" — all waves walk their
// 16 KiB tiles in lockstep (row k of every tile at about the same time), so address bits [13:10] are common to all
// in-flight stores and only the hash with the upper (placement-dependent) bits spreads them over the HBM channels.
// Variants of the same pure-write sweep on hipMalloc(3 GB) buffers, fast and slow ones:
//   V0 rows in order                       V1 each wave starts at its own row (k + hash(wave)) & 15
//   V2 each wave walks rows in a wave-specific permutation (k * odd + r) & 15
//   V3 V1 with 64 KiB tiles                V4 rows in order, tile order bit-reversed within groups of 16 tiles
// and a phase scan: 1 GB windows at 32 MB steps through one buffer for V0 and V1.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

__device__ __forceinline__ unsigned mixhash(unsigned x) { x ^= x >> 7; x *= 0x9E3779B1u; x ^= x >> 15; return x; }

template <int V>
__global__ __launch_bounds__(256) void wr(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    if (V == 3) {
        const long n64 = n_tiles / 4;
        const unsigned r = mixhash((unsigned)gw) & 63;
        for (long t = gw; t < n64; t += nw) {
            f64x2* d = dst + t * (64 * 64);
#pragma unroll 16
            for (int k = 0; k < 64; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + ((k + r) & 63) * 64 + lane);
        }
        return;
    }
    const unsigned h = mixhash((unsigned)gw);
    const unsigned r = h & 15, m = ((h >> 4) & 7) * 2 + 1;
    for (long t = gw; t < n_tiles; t += nw) {
        long tile = t;
        if (V == 4) { const long lo = t & 15; tile = (t & ~15L) | (long)(((lo & 1) << 3) | ((lo & 2) << 1) | ((lo & 4) >> 1) | ((lo & 8) >> 3)); }
        f64x2* d = dst + tile * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int row = k;
            if (V == 1) row = (k + r) & 15;
            if (V == 2) row = (k * m + r) & 15;
            __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + row * 64 + lane);
        }
    }
}
static hipStream_t st;
float timeit(int launches, int reps, const std::function<void()>& fn) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> v;
    fn();
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, st)); for (int l = 0; l < launches; ++l) fn(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms / launches);
    }
    std::sort(v.begin(), v.end()); CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[v.size() / 2];
}
template <int V>
double bw(void* p, size_t bytes, int grid = 4096) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(wr<V>, dim3(grid), dim3(256), 0, st, n, (f64x2*)p); });
    return bytes / ms / 1e6;
}

int main() {
    CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, GB = 1ull << 30, B3 = 3 * GB;
    std::vector<char*> bufs;
    std::vector<double> w;
    for (int i = 0; i < 12; ++i) {
        char* b; CK(hipMalloc(&b, B3 + 1 * GB));
        bufs.push_back(b);
        w.push_back(bw<0>(b, B3));
        printf("buf %2d va %p: V0 %5.0f  V1 %5.0f  V2 %5.0f  V3 %5.0f  V4 %5.0f | grid 2048: V0 %5.0f V1 %5.0f | grid 8192: V0 %5.0f V1 %5.0f\n", i, (void*)b, w.back(),
               bw<1>(b, B3), bw<2>(b, B3), bw<3>(b, B3), bw<4>(b, B3), bw<0>(b, B3, 2048), bw<1>(b, B3, 2048), bw<0>(b, B3, 8192), bw<1>(b, B3, 8192));
        fflush(stdout);
    }
    const int ifast = (int)(std::max_element(w.begin(), w.end()) - w.begin());
    const int islow = (int)(std::min_element(w.begin(), w.end()) - w.begin());
    for (int which : {ifast, islow}) {
        printf("phase scan buf %d (%s), 1 GB windows at 32 MB steps, V0:", which, which == ifast ? "fastest" : "slowest");
        for (size_t off = 0; off <= 3 * GB; off += 32 * MB) printf(" %.0f", bw<0>(bufs[which] + off, GB));
        printf("\nphase scan buf %d, V1:", which);
        for (size_t off = 0; off <= 3 * GB; off += 32 * MB) printf(" %.0f", bw<1>(bufs[which] + off, GB));
        printf("\n");
        fflush(stdout);
    }
    // sweep length: how long must a sweep be for slow/fast to show?
    for (int which : {ifast, islow}) {
        printf("sweep length buf %d V0:", which);
        for (size_t len : {256 * MB, 512 * MB, 1 * GB, 2 * GB, 3 * GB, 4 * GB}) printf("  %zu MB: %.0f", len / MB, bw<0>(bufs[which], len));
        printf("\n");
    }
    return 0;
}
