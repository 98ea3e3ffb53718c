// place_exp.hip — what makes streaming-WRITE bandwidth bimodal between hipMalloc allocations on MI355X?
// (round-1 map: profiles/r01_hbm_placement_map.txt.)  Phases:
//   P1  16 consecutive hipMalloc(3 GB): write / read bandwidth of each (the round-1 map, shorter)
//   P2  fastest and slowest of them: (a) bandwidth of every 256 MB sub-range, (b) page-touch rate at 4 KiB / 64 KiB /
//       2 MiB stride (one 16-byte store per lane per page: pure address-translation stress, hardly any bytes)
//   P3  virtual-memory API: granularities; physical chunks of 256 MB made with hipMemCreate and mapped side by side:
//       bandwidth per chunk; then the slowest and the fastest chunk are mapped a second time at OTHER virtual
//       addresses (same physical memory, new VA) -> is "slow" a property of the physical chunk or of the mapping?
//   P4  one hipMalloc of 36 GB: bandwidth per 3 GB piece and per 256 MB piece
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)
#define TRY(x) ([&]() { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  [%s failed: %s]\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); return false; } return true; }())

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
// one 16-byte store per lane, lanes `stride` bytes apart: n_pages stores in total
__global__ __launch_bounds__(256) void touch(long n_pages, long stride, char* __restrict__ dst, int rep) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_pages; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(f64x2{(double)i, (double)rep}, reinterpret_cast<f64x2*>(dst + i * stride + (rep & 15) * 16));
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
double wr_bw(void* p, size_t bytes, int launches = 4, int reps = 3) {
    const long nt = bytes / 16384;
    float ms = timeit(launches, reps, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, nt, (f64x2*)p); });
    return bytes / ms / 1e6;
}
double rd_bw(void* p, size_t bytes, f64x2* sink) {
    const long nt = bytes / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, st, nt, (const f64x2*)p, sink); });
    return bytes / ms / 1e6;
}
double touch_rate(void* p, size_t bytes, long stride) {   // M pages per second
    const long n = bytes / stride;
    int rep = 0;
    long blocks = std::min<long>((n + 255) / 256, 8192);
    float ms = timeit(8, 3, [&] { hipLaunchKernelGGL(touch, dim3((int)blocks), dim3(256), 0, st, n, stride, (char*)p, rep++); });
    return n / ms / 1e3;
}

int main(int argc, char** argv) {
    CK(hipStreamCreate(&st));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("P0 free %.1f GB of %.1f GB\n", free_b / 1e9, total_b / 1e9);
    f64x2* sink; CK(hipMalloc(&sink, 4096));
    const size_t MB = 1ull << 20;
    const size_t B3 = 3072 * MB, SUB = 256 * MB;

    // ---------------- P1
    const int nbuf = 16;
    std::vector<char*> bufs(nbuf);
    std::vector<double> w(nbuf);
    for (auto& b : bufs) CK(hipMalloc(&b, B3));
    for (int i = 0; i < nbuf; ++i) {
        w[i] = wr_bw(bufs[i], B3);
        double r = rd_bw(bufs[i], B3, sink);
        printf("P1 buf %2d va %p write %7.1f read %7.1f GB/s\n", i, (void*)bufs[i], w[i], r);
    }
    fflush(stdout);
    const int ifast = (int)(std::max_element(w.begin(), w.end()) - w.begin());
    const int islow = (int)(std::min_element(w.begin(), w.end()) - w.begin());

    // ---------------- P2
    for (int which : {ifast, islow}) {
        printf("P2 buf %d (%s, %.0f GB/s): write GB/s per 256 MB sub-range:", which, which == ifast ? "fastest" : "slowest", w[which]);
        for (size_t off = 0; off < B3; off += SUB) printf(" %.0f", wr_bw(bufs[which] + off, SUB, 8, 3));
        printf("\n");
        printf("P2 buf %d page-touch rate (M stores/s, one 16-byte store per page): 4 KiB %.1f | 64 KiB %.1f | 2 MiB %.2f\n", which,
               touch_rate(bufs[which], B3, 4096), touch_rate(bufs[which], B3, 65536), touch_rate(bufs[which], B3, 2 * MB));
    }
    fflush(stdout);
    for (auto& b : bufs) CK(hipFree(b));

    // ---------------- P3
    int dev = 0; CK(hipGetDevice(&dev));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    if (TRY(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum)) &&
        TRY(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended))) {
        printf("P3 VMM granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
        const int nchunk = 48;
        const size_t CH = SUB;
        char* va = nullptr;
        if (TRY(hipMemAddressReserve((void**)&va, CH * (nchunk + 4), 2 * MB, nullptr, 0))) {
            std::vector<hipMemGenericAllocationHandle_t> h(nchunk);
            std::vector<double> cw(nchunk, 0.0);
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            int made = 0;
            for (int i = 0; i < nchunk; ++i) {
                if (!TRY(hipMemCreate(&h[i], CH, &prop, 0))) break;
                if (!TRY(hipMemMap(va + CH * i, CH, 0, h[i], 0))) break;
                if (!TRY(hipMemSetAccess(va + CH * i, CH, &acc, 1))) break;
                ++made;
            }
            printf("P3 %d physical chunks of 256 MB mapped at %p; write GB/s per chunk:", made, (void*)va);
            for (int i = 0; i < made; ++i) { cw[i] = wr_bw(va + CH * i, CH, 8, 3); printf(" %.0f", cw[i]); }
            printf("\n");
            // 3 GB windows over 12 consecutive chunks (what a kernel would see)
            for (int i = 0; i + 12 <= made; i += 12) printf("P3 window chunks %d..%d (3 GB) write %.1f GB/s\n", i, i + 11, wr_bw(va + CH * i, 12 * CH));
            if (made >= 2) {
                const int cf = (int)(std::max_element(cw.begin(), cw.begin() + made) - cw.begin());
                const int cs = (int)(std::min_element(cw.begin(), cw.begin() + made) - cw.begin());
                // second mapping of the same physical chunks at fresh virtual addresses
                char* va2 = va + CH * nchunk;
                if (TRY(hipMemMap(va2, CH, 0, h[cs], 0)) && TRY(hipMemSetAccess(va2, CH, &acc, 1)) &&
                    TRY(hipMemMap(va2 + CH, CH, 0, h[cf], 0)) && TRY(hipMemSetAccess(va2 + CH, CH, &acc, 1))) {
                    printf("P3 slowest chunk %d: %.0f GB/s at its first VA, %.0f GB/s through a second mapping; fastest chunk %d: %.0f / %.0f\n",
                           cs, wr_bw(va + CH * cs, CH, 8, 3), wr_bw(va2, CH, 8, 3), cf, wr_bw(va + CH * cf, CH, 8, 3), wr_bw(va2 + CH, CH, 8, 3));
                    (void)hipMemUnmap(va2, CH);
                    (void)hipMemUnmap(va2 + CH, CH);
                }
                // a 3 GB virtual slab stitched from the 12 FASTEST chunks vs from the 12 SLOWEST
                if (made >= 24) {
                    std::vector<int> order(made);
                    for (int i = 0; i < made; ++i) order[i] = i;
                    std::sort(order.begin(), order.end(), [&](int a, int b) { return cw[a] > cw[b]; });
                    for (int i = 0; i < made; ++i) (void)hipMemUnmap(va + CH * i, CH);
                    bool ok = true;
                    for (int k = 0; k < 12 && ok; ++k) ok = TRY(hipMemMap(va + CH * k, CH, 0, h[order[k]], 0));
                    for (int k = 0; k < 12 && ok; ++k) ok = TRY(hipMemMap(va + CH * (12 + k), CH, 0, h[order[made - 1 - k]], 0));
                    if (ok && TRY(hipMemSetAccess(va, 24 * CH, &acc, 1)))
                        printf("P3 3 GB slab stitched from the 12 fastest chunks: %.1f GB/s; from the 12 slowest: %.1f GB/s\n",
                               wr_bw(va, 12 * CH), wr_bw(va + 12 * CH, 12 * CH));
                    for (int k = 0; k < 24; ++k) (void)hipMemUnmap(va + CH * k, CH);
                } else {
                    for (int i = 0; i < made; ++i) (void)hipMemUnmap(va + CH * i, CH);
                }
            }
            for (int i = 0; i < made; ++i) (void)hipMemRelease(h[i]);
            (void)hipMemAddressFree(va, CH * (nchunk + 4));
        }
    }
    fflush(stdout);

    // ---------------- P4
    const int npiece = 12;
    char* big = nullptr;
    if (TRY(hipMalloc(&big, B3 * npiece))) {
        printf("P4 one hipMalloc of %.0f GB at %p; write GB/s per 3 GB piece:", B3 * npiece / 1e9, (void*)big);
        for (int i = 0; i < npiece; ++i) printf(" %.0f", wr_bw(big + B3 * i, B3));
        printf("\nP4 per 256 MB piece:");
        for (size_t off = 0; off < B3 * npiece; off += SUB) printf(" %.0f", wr_bw(big + off, SUB, 8, 3));
        printf("\n");
        CK(hipFree(big));
    }
    return 0;
}
