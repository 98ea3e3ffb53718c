// place_exp3.hip — is streaming-write speed a property of PHYSICAL memory at ~128 MB granularity, and can a fast
// virtual slab be stitched from fast physical chunks?  (place_exp.hip measured 256 MB pieces one at a time; a 256 MB
// working set sits in the 256 MB Infinity Cache, so those per-piece numbers said nothing about HBM.)
//   1. hipMemCreate NCH physical chunks of 128 MB, map them side by side.
//   2. write bandwidth of every 1 GB window (8 chunks), sliding by one chunk -> a map with 128 MB resolution.
//   3. chunk score = mean of the windows that contain it; stitch 3 GB slabs from the best-scored chunks, the
//      worst-scored chunks, and alternating best/worst; measure write, read and the 13:43 mix.
//   4. release everything, create the chunks again, measure the window map again (does the pattern follow the
//      allocation order = physical addresses?).
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
double rd_bw(void* p, size_t bytes, f64x2* sink) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(rd, dim3(4096), dim3(256), 0, st, n, (const f64x2*)p, sink); });
    return bytes / ms / 1e6;
}
double mix_bw(void* src, void* dst, size_t dst_bytes) {
    const long n_tiles = dst_bytes / (43 * 1024);
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(4096), dim3(256), 0, st, n_tiles, (const f64x2*)src, (f64x2*)dst); });
    return n_tiles * 56.0 * 1024 / ms / 1e6;
}

int main(int argc, char** argv) {
    const int NCH = argc > 1 ? atoi(argv[1]) : 384;   // 384 x 128 MB = 48 GB
    CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, CH = 128 * MB;
    const int W = 8;   // chunks per window
    int dev = 0; CK(hipGetDevice(&dev));
    f64x2* sink; CK(hipMalloc(&sink, 4096));
    char* src; CK(hipMalloc(&src, 1024 * MB));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char* va = nullptr;
    CK(hipMemAddressReserve((void**)&va, CH * (size_t)(NCH + 96), 2 * MB, nullptr, 0));

    for (int round = 0; round < 2; ++round) {
        std::vector<hipMemGenericAllocationHandle_t> h(NCH);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < NCH; ++i) {
            CK(hipMemCreate(&h[i], CH, &prop, 0));
            CK(hipMemMap(va + CH * i, CH, 0, h[i], 0));
        }
        CK(hipMemSetAccess(va, CH * (size_t)NCH, &acc, 1));
        std::vector<double> win(NCH - W + 1);
        printf("round %d: write GB/s of the 1 GB window starting at chunk i (128 MB chunks, %d chunks):\n", round, NCH);
        for (int i = 0; i + W <= NCH; ++i) {
            win[i] = wr_bw(va + CH * i, CH * W);
            printf(" %.0f", win[i]);
            if (i % 32 == 31) printf("\n");
        }
        printf("\n");
        fflush(stdout);
        if (round == 0) {
            std::vector<double> score(NCH, 0.0);
            for (int c = 0; c < NCH; ++c) {
                int cnt = 0;
                for (int i = std::max(0, c - W + 1); i <= std::min(c, NCH - W); ++i) { score[c] += win[i]; ++cnt; }
                score[c] /= cnt;
            }
            std::vector<int> order(NCH);
            std::iota(order.begin(), order.end(), 0);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return score[a] > score[b]; });
            printf("chunk scores: best %.0f, 24th %.0f, median %.0f, worst %.0f\n", score[order[0]], score[order[23]], score[order[NCH / 2]], score[order[NCH - 1]]);
            // second mappings of chosen chunks behind the first ones (a handle may be mapped more than once)
            char* v2 = va + CH * (size_t)NCH;
            auto stitch = [&](const std::vector<int>& ids, const char* name) {
                for (size_t k = 0; k < ids.size(); ++k) CK(hipMemMap(v2 + CH * k, CH, 0, h[ids[k]], 0));
                CK(hipMemSetAccess(v2, CH * ids.size(), &acc, 1));
                const size_t bytes = CH * ids.size();
                printf("stitched %-28s (%zu chunks): write %.0f  read %.0f  mix13:43 %.0f GB/s\n", name, ids.size(), wr_bw(v2, bytes), rd_bw(v2, bytes, sink),
                       mix_bw(src, v2, bytes));
                for (size_t k = 0; k < ids.size(); ++k) CK(hipMemUnmap(v2 + CH * k, CH));
                fflush(stdout);
            };
            std::vector<int> best(order.begin(), order.begin() + 24), worst(order.end() - 24, order.end()), alt, best_sorted(best), shuffled(best);
            for (int k = 0; k < 12; ++k) { alt.push_back(order[k]); alt.push_back(order[NCH - 1 - k]); }
            std::sort(best_sorted.begin(), best_sorted.end());
            stitch(best, "24 best, by score");
            stitch(best_sorted, "24 best, in address order");
            stitch(worst, "24 worst");
            stitch(alt, "12 best + 12 worst alternating");
            std::vector<int> mid(order.begin() + NCH / 2 - 12, order.begin() + NCH / 2 + 12);
            stitch(mid, "24 around the median");
            std::vector<int> first24(24);
            std::iota(first24.begin(), first24.end(), 0);
            stitch(first24, "chunks 0..23 (plain)");
        }
        for (int i = 0; i < NCH; ++i) { CK(hipMemUnmap(va + CH * i, CH)); CK(hipMemRelease(h[i])); }
    }
    return 0;
}
