// place_exp2.hip — follow-up to place_exp.hip: a "fast" 3 GB buffer is fast only as a whole sweep, its 256 MB pieces are not.
// Hypothesis A (XCD <-> HBM-stack affinity): with a fixed workgroup -> address map every XCD writes a fixed residue class
//   of addresses; whether that class sits on near or far HBM stacks depends on the physical phase of the allocation.
//   Test: partition the buffer into classes by (addr / G) % 8 and let XCD x write class (x + phase) % 8; sweep G and phase
//   on a fast and on a slow buffer.
// Hypothesis B (concurrency window): vary the persistent grid size.
// Also: plain vs non-temporal stores, and a read/write mix like the von Mises kernel's (13 : 43).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); fflush(stdout); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

template <bool NT>
__global__ __launch_bounds__(256) void wr(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long t = (long)blockIdx.x * 4 + wave; t < n_tiles; t += (long)gridDim.x * 4) {
        f64x2* d = dst + t * (16 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (NT) __builtin_nontemporal_store(f64x2{(double)t, (double)k}, d + k * 64 + lane);
            else d[k * 64 + lane] = f64x2{(double)t, (double)k};
        }
    }
}

// rows of 1 KiB; row r belongs to unit r >> lg (unit = G bytes = 2^lg rows), class = unit & 7. XCD x writes the rows of
// class (x + phase) & 7, sixteen consecutive class-rows per wave iteration.
__global__ __launch_bounds__(256) void wr_aff(long rows_per_class, int lg, int phase, int use_hw_id, f64x2* __restrict__ dst,
                                              unsigned* __restrict__ mismatch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hw = xcc_id();
    const int x = use_hw_id ? hw : (int)(blockIdx.x & 7);
    if (mismatch && lane == 0 && wave == 0 && hw != (int)(blockIdx.x & 7)) atomicAdd(mismatch, 1u);
    const int cls = (x + phase) & 7;
    const long lw = (long)(blockIdx.x >> 3) * 4 + wave, nw = (long)(gridDim.x >> 3) * 4;
    const long rpu = 1L << lg;
    for (long it = lw; it * 16 < rows_per_class; it += nw) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const long q = it * 16 + k;
            if (q < rows_per_class) {
                const long unit = (q >> lg) * 8 + cls;
                const long row = unit * rpu + (q & (rpu - 1));
                __builtin_nontemporal_store(f64x2{(double)q, (double)k}, dst + row * 64 + lane);
            }
        }
    }
}

// von-Mises-like mix: per tile of 64 points read 13 KiB (two 6 KiB blocks + 0.5 KiB... rounded: 13 rows) and write 43 rows
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
double wr_bw(void* p, size_t bytes, int grid = 4096, bool nt = true) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] {
        if (nt) hipLaunchKernelGGL(wr<true>, dim3(grid), dim3(256), 0, st, n, (f64x2*)p);
        else hipLaunchKernelGGL(wr<false>, dim3(grid), dim3(256), 0, st, n, (f64x2*)p);
    });
    return bytes / ms / 1e6;
}

int main() {
    CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, B3 = 3072 * MB;
    std::vector<char*> bufs;
    std::vector<double> w;
    int ifast = -1, islow = -1;
    for (int i = 0; i < 40; ++i) {
        char* b; CK(hipMalloc(&b, B3));
        bufs.push_back(b);
        w.push_back(wr_bw(b, B3));
        printf("buf %2d write %7.1f GB/s\n", i, w.back());
        if (w.back() > 6500 && ifast < 0) ifast = i;
        if (w.back() < 6100 && islow < 0) islow = i;
        if (ifast >= 0 && islow >= 0 && i >= 7) break;
    }
    if (ifast < 0) { ifast = (int)(std::max_element(w.begin(), w.end()) - w.begin()); printf("no fast buffer found; using the best\n"); }
    if (islow < 0) islow = (int)(std::min_element(w.begin(), w.end()) - w.begin());
    printf("fast = buf %d (%.0f), slow = buf %d (%.0f)\n", ifast, w[ifast], islow, w[islow]);
    fflush(stdout);

    unsigned* mism; CK(hipMalloc(&mism, 4)); CK(hipMemset(mism, 0, 4));
    for (int which : {ifast, islow}) {
        char* p = bufs[which];
        printf("== buf %d (%s)\n", which, which == ifast ? "fast" : "slow");
        printf("E1 persistent grid size, NT stores:");
        for (int g : {256, 512, 1024, 2048, 3072, 4096, 8192, 16384}) printf("  %d: %.0f", g, wr_bw(p, B3, g));
        printf("\nE1 plain stores:");
        for (int g : {2048, 4096}) printf("  %d: %.0f", g, wr_bw(p, B3, g, false));
        printf("\nE1 one tile per wave (grid = tiles/4): %.0f\n", wr_bw(p, B3, (int)(B3 / 16384 / 4)));
        const long rows = B3 / 1024, rpc = rows / 8;
        for (int lg = 0; lg <= 15; ++lg) {   // G = 1 KiB .. 32 MiB
            printf("E2 G = %6ld KiB, phase 0..7:", 1L << lg);
            double lo = 1e30, hi = 0;
            for (int ph = 0; ph < 8; ++ph) {
                float ms = timeit(3, 3, [&] { hipLaunchKernelGGL(wr_aff, dim3(4096), dim3(256), 0, st, rpc, lg, ph, 1, (f64x2*)p, mism); });
                const double bw = B3 / ms / 1e6;
                lo = std::min(lo, bw); hi = std::max(hi, bw);
                printf(" %5.0f", bw);
            }
            printf("   spread %.1f %%\n", (hi - lo) / hi * 100);
            fflush(stdout);
        }
    }
    unsigned hm = 0; CK(hipMemcpy(&hm, mism, 4, hipMemcpyDeviceToHost));
    printf("workgroups whose HW_REG_XCC_ID != blockIdx %% 8: %u\n", hm);

    // E3: von Mises mix, inputs from the slow buffer's neighbour, outputs into fast / slow
    {
        const long n_tiles = B3 / (43 * 1024);
        char* src = bufs[(islow + 1) % bufs.size() == (size_t)ifast ? (islow + 2) % bufs.size() : (islow + 1) % bufs.size()];
        for (int which : {ifast, islow}) {
            for (int g : {2048, 4096, (int)((n_tiles + 3) / 4)}) {
                float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(g), dim3(256), 0, st, n_tiles, (const f64x2*)src, (f64x2*)bufs[which]); });
                printf("E3 mix 13:43 into buf %d grid %d: %.0f GB/s\n", which, g, n_tiles * 56.0 * 1024 / ms / 1e6);
            }
        }
    }
    return 0;
}
