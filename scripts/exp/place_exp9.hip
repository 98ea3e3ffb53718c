// place_exp9.hip — which ALLOCATION STRATEGY yields a fast output slab, and does the class survive once the helper
// mappings are gone?  All rates are 3.44 GB streaming-write sweeps (the bench's output block) and the 13:43 mix.
//   S1  8 x hipMalloc(3.44 GB), coexisting                       (round 1's bench: keep the fastest)
//   S2  ONE physical block mapped at 12 ranges one after the other (dxo_output_alloc as first written)
//   S3  12 physical blocks mapped side by side                    (place_exp7 pattern)
//   S4  with S3's blocks still mapped: a second mapping of block 0 at a fresh range; then S3's other blocks are
//       unmapped and released and the rate of that mapping is measured again
//   S5  "ballast": a 1 GB physical block mapped 40 times (40 GB of address space, 1 GB of memory), then the real
//       block at a fresh range behind it; measured; ballast unmapped; measured again; then 8 further fresh ranges
//       one after the other with the ballast gone
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
        f64x2* d = dst + t * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
    }
}
__global__ __launch_bounds__(256) void mix(long n_tiles, const f64x2* __restrict__ src, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        const f64x2* s = src + t * (13 * 64);
        f64x2* d = dst + t * (43 * 64);
        f64x2 acc = {0, 0};
#pragma unroll
        for (int k = 0; k < 13; ++k) acc += s[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 43; ++k) __builtin_nontemporal_store(acc + f64x2{(double)k, 0.0}, d + k * 64 + lane);
    }
}
static hipStream_t st;
static char* g_src;
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
const size_t MB = 1ull << 20, GB = 1ull << 30;
const size_t SLAB = 3520 * MB;   // >= 10^7 points x 344 B
double wr_bw(void* p) {
    const long n = SLAB / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, n, (f64x2*)p); });
    return n * 16384.0 / ms / 1e6;
}
double mix_bw(void* dst) {
    const long n_tiles = SLAB / (43 * 1024);
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(4096), dim3(256), 0, st, n_tiles, (const f64x2*)g_src, (f64x2*)dst); });
    return n_tiles * 56.0 * 1024 / ms / 1e6;
}
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
hipMemGenericAllocationHandle_t mk(size_t bytes) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, bytes, &prop, 0)); return h; }
void map_at(hipMemGenericAllocationHandle_t h, char* va, size_t bytes) { CK(hipMemMap(va, bytes, 0, h, 0)); CK(hipMemSetAccess(va, bytes, &acc, 1)); }
void rep(const char* what, char* va) { printf("%-74s write %5.0f  mix %5.0f GB/s\n", what, wr_bw(va), mix_bw(va)); fflush(stdout); }

int main() {
    CK(hipStreamCreate(&st));
    int dev = 0; CK(hipGetDevice(&dev));
    CK(hipMalloc(&g_src, 1100 * MB));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char name[128];
    {   // S1
        std::vector<char*> b(8);
        for (auto& x : b) CK(hipMalloc(&x, SLAB));
        for (int i = 0; i < 8; ++i) { snprintf(name, sizeof name, "S1 hipMalloc %d", i); rep(name, b[i]); }
        for (auto& x : b) CK(hipFree(x));
    }
    const size_t STRIDE = 4 * GB;
    {   // S2
        char* R; CK(hipMemAddressReserve((void**)&R, 12 * STRIDE, 2 * MB, nullptr, 0));
        auto h = mk(SLAB);
        for (int k = 0; k < 12; ++k) {
            map_at(h, R + k * STRIDE, SLAB);
            snprintf(name, sizeof name, "S2 one block, range %d (one at a time)", k); rep(name, R + k * STRIDE);
            CK(hipMemUnmap(R + k * STRIDE, SLAB));
        }
        CK(hipMemRelease(h)); CK(hipMemAddressFree(R, 12 * STRIDE));
    }
    {   // S3 + S4
        const int NS = 12;
        char* R; CK(hipMemAddressReserve((void**)&R, (NS + 2) * STRIDE, 2 * MB, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> h(NS);
        for (int k = 0; k < NS; ++k) { h[k] = mk(SLAB); map_at(h[k], R + k * STRIDE, SLAB); }
        for (int k = 0; k < NS; ++k) { snprintf(name, sizeof name, "S3 block %d of %d mapped side by side", k, NS); rep(name, R + k * STRIDE); }
        char* V = R + NS * STRIDE;
        map_at(h[0], V, SLAB);
        rep("S4 second mapping of block 0 at a fresh range, others still mapped", V);
        for (int k = 1; k < NS; ++k) { CK(hipMemUnmap(R + k * STRIDE, SLAB)); CK(hipMemRelease(h[k])); }
        rep("S4 the same mapping after the other blocks were unmapped and released", V);
        rep("S4 block 0's FIRST mapping now", R);
        CK(hipMemUnmap(V, SLAB)); CK(hipMemUnmap(R, SLAB)); CK(hipMemRelease(h[0])); CK(hipMemAddressFree(R, (NS + 2) * STRIDE));
    }
    {   // S5
        const int NB = 40;
        char* R; CK(hipMemAddressReserve((void**)&R, (size_t)NB * GB + 10 * STRIDE, 2 * MB, nullptr, 0));
        auto ballast = mk(GB);
        auto h = mk(SLAB);
        char* V = R + (size_t)NB * GB;
        map_at(h, V, SLAB);
        rep("S5 real block before any ballast", V);
        CK(hipMemUnmap(V, SLAB));
        for (int k = 0; k < NB; ++k) map_at(ballast, R + (size_t)k * GB, GB);
        map_at(h, V, SLAB);
        rep("S5 real block behind 40 GB of ballast mappings (1 GB of memory)", V);
        for (int k = 0; k < NB; ++k) CK(hipMemUnmap(R + (size_t)k * GB, GB));
        rep("S5 the same mapping, ballast unmapped", V);
        CK(hipMemUnmap(V, SLAB));
        for (int k = 1; k <= 8; ++k) {
            map_at(h, V + k * STRIDE, SLAB);
            snprintf(name, sizeof name, "S5 fresh range %d after the ballast is gone (one at a time)", k); rep(name, V + k * STRIDE);
            CK(hipMemUnmap(V + k * STRIDE, SLAB));
        }
        // ballast of DISTINCT physical memory instead of aliases
        std::vector<hipMemGenericAllocationHandle_t> hb(NB);
        for (int k = 0; k < NB; ++k) { hb[k] = mk(GB); map_at(hb[k], R + (size_t)k * GB, GB); }
        map_at(h, V, SLAB);
        rep("S5 real block behind 40 GB of DISTINCT ballast memory", V);
        for (int k = 0; k < NB; ++k) { CK(hipMemUnmap(R + (size_t)k * GB, GB)); CK(hipMemRelease(hb[k])); }
        rep("S5 the same mapping, distinct ballast released", V);
    }
    return 0;
}
