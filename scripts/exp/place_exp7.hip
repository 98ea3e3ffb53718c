// place_exp7.hip — only sweeps of >= 3 GB are trusted from here on (a 1 GB window repeated back to back interacts with
// the 256 MB Infinity Cache and has a bimodality of its own: place_exp4 f shows a slow 1 GB window inside a fast 3 GB
// window). Hypothesis left standing: write speed is a property of PHYSICAL memory at multi-GB scale.
//   1. NS slabs of 3 GB, each 24 x 128 MB physical chunks (hipMemCreate), mapped side by side: 3 GB sweep of each.
//   2. swap test: the chunks of the fastest slab mapped at the slowest slab's virtual address and vice versa.
//   3. hybrids at a fresh virtual address: chunks drawn from two fast slabs (interleaved), from two slow slabs, half/half.
//   4. aliasing check: a second mapping really is the same memory.
//   5. map of the whole area with 3 GB sweeps at 1 GB steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
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
double wr_bw(void* p, size_t bytes) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, n, (f64x2*)p); });
    return bytes / ms / 1e6;
}
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
typedef std::vector<hipMemGenericAllocationHandle_t> Chunks;
const size_t MB = 1ull << 20, GB = 1ull << 30, CH = 128 * MB, B3 = 3 * GB;
void map_at(const Chunks& h, char* va) {
    for (size_t i = 0; i < h.size(); ++i) CK(hipMemMap(va + CH * i, CH, 0, h[i], 0));
    CK(hipMemSetAccess(va, CH * h.size(), &acc, 1));
}
void unmap_at(const Chunks& h, char* va) { for (size_t i = 0; i < h.size(); ++i) CK(hipMemUnmap(va + CH * i, CH)); }

int main(int argc, char** argv) {
    const int NS = argc > 1 ? atoi(argv[1]) : 20;
    CK(hipStreamCreate(&st));
    int dev = 0; CK(hipGetDevice(&dev));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char* R = nullptr;
    CK(hipMemAddressReserve((void**)&R, (size_t)(NS + 4) * B3, 2 * MB, nullptr, 0));
    std::vector<Chunks> slab(NS, Chunks(24));
    std::vector<double> w(NS);
    for (int s = 0; s < NS; ++s) {
        for (auto& x : slab[s]) CK(hipMemCreate(&x, CH, &prop, 0));
        map_at(slab[s], R + (size_t)s * B3);
    }
    printf("1. 3 GB sweeps of %d slabs (24 x 128 MB chunks each) at %p:\n", NS, (void*)R);
    for (int s = 0; s < NS; ++s) { w[s] = wr_bw(R + (size_t)s * B3, B3); printf(" %.0f", w[s]); }
    printf("\n   again:");
    for (int s = 0; s < NS; ++s) printf(" %.0f", wr_bw(R + (size_t)s * B3, B3));
    printf("\n");
    fflush(stdout);
    std::vector<int> order(NS);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    const int f0 = order[0], f1 = order[1], s0 = order[NS - 1], s1 = order[NS - 2];
    printf("fast slabs %d (%.0f), %d (%.0f); slow slabs %d (%.0f), %d (%.0f)\n", f0, w[f0], f1, w[f1], s0, w[s0], s1, w[s1]);
    // 2. swap
    unmap_at(slab[f0], R + (size_t)f0 * B3);
    unmap_at(slab[s0], R + (size_t)s0 * B3);
    map_at(slab[f0], R + (size_t)s0 * B3);
    map_at(slab[s0], R + (size_t)f0 * B3);
    printf("2. swapped: fast slab's chunks at the slow slab's address: %.0f; slow slab's chunks at the fast slab's address: %.0f GB/s\n",
           wr_bw(R + (size_t)s0 * B3, B3), wr_bw(R + (size_t)f0 * B3, B3));
    unmap_at(slab[f0], R + (size_t)s0 * B3);
    unmap_at(slab[s0], R + (size_t)f0 * B3);
    map_at(slab[f0], R + (size_t)f0 * B3);
    map_at(slab[s0], R + (size_t)s0 * B3);
    printf("   back in place: %.0f / %.0f\n", wr_bw(R + (size_t)f0 * B3, B3), wr_bw(R + (size_t)s0 * B3, B3));
    fflush(stdout);
    // 3. hybrids at a fresh address (second mappings)
    char* V = R + (size_t)NS * B3;
    auto hybrid = [&](const char* name, int a, int b, int mode) {
        Chunks h(24);
        for (int k = 0; k < 24; ++k) {
            if (mode == 0) h[k] = (k & 1) ? slab[b][k] : slab[a][k];           // alternate chunk by chunk
            else if (mode == 1) h[k] = k < 12 ? slab[a][k] : slab[b][k];        // first half a, second half b
            else h[k] = slab[a][23 - k];                                       // a reversed
        }
        map_at(h, V);
        printf("3. %-52s %.0f GB/s\n", name, wr_bw(V, B3));
        unmap_at(h, V);
        fflush(stdout);
    };
    hybrid("fast+fast alternating", f0, f1, 0);
    hybrid("slow+slow alternating", s0, s1, 0);
    hybrid("fast+slow alternating", f0, s0, 0);
    hybrid("fast first half, slow second half", f0, s0, 1);
    hybrid("fast slab reversed", f0, f0, 2);
    hybrid("slow slab reversed", s0, s0, 2);
    hybrid("fast slab as it is (second mapping)", f0, f0, 1);
    hybrid("slow slab as it is (second mapping)", s0, s0, 1);
    // 4. aliasing check
    {
        map_at(slab[f0], V);
        double probe = 1234.5;
        CK(hipMemcpy(R + (size_t)f0 * B3 + 4096, &probe, 8, hipMemcpyHostToDevice));
        double back = 0;
        CK(hipMemcpy(&back, V + 4096, 8, hipMemcpyDeviceToHost));
        printf("4. second mapping aliases the first: %s\n", back == probe ? "yes" : "NO");
        unmap_at(slab[f0], V);
    }
    // 5. 3 GB sweeps at 1 GB steps over the whole area
    printf("5. 3 GB sweeps at 1 GB steps:");
    for (size_t off = 0; off + B3 <= (size_t)NS * B3; off += GB) printf(" %.0f", wr_bw(R + off, B3));
    printf("\n");
    return 0;
}
