// place_exp5.hip — model after place_exp3/4: streaming-write speed is a property of the 1 GB-aligned VIRTUAL block
// (= one page-directory page of 512 x 2 MB entries), fixed when the block is first mapped, whatever physical memory
// is mapped there. Tests:
//   1. ONE physical slab of 1 GB (8 x 128 MB) mapped, measured and unmapped at each of 48 consecutive 1 GB-aligned
//      virtual blocks -> bimodal map with identical physical memory?  2. second pass: reproducible?
//   3. a fast and a slow block: XCD-partitioned sweep (every XCD writes its own contiguous eighth, 8x fewer distinct
//      pages per XCD) vs the interleaved sweep.
//   4. DIFFERENT physical memory at the same two blocks.
//   5. offsets inside a block: window of 1 GB starting at +0, +256, +512, +768 MB across a fast/slow border.
//   6. a second reservation made after freeing the first: same virtual address? same map?
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
// every XCD (blockIdx % 8, the observed dispatch rule) sweeps its own contiguous eighth of the buffer
__global__ __launch_bounds__(256) void wr_part(long n_tiles, f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per = n_tiles / 8;
    const long x = blockIdx.x & 7, lw = (long)(blockIdx.x >> 3) * 4 + wave, nw = (long)(gridDim.x >> 3) * 4;
    for (long t = lw; t < per; t += nw) {
        f64x2* d = dst + (x * per + t) * (16 * 64);
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
double wr_bw(void* p, size_t bytes, bool part = false) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] {
        if (part) hipLaunchKernelGGL(wr_part, dim3(4096), dim3(256), 0, st, n, (f64x2*)p);
        else hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, n, (f64x2*)p);
    });
    return bytes / ms / 1e6;
}
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
struct Slab { std::vector<hipMemGenericAllocationHandle_t> h; size_t ch = 0; size_t bytes() const { return ch * h.size(); } };
Slab make(size_t total, size_t ch) {
    Slab s; s.ch = ch; s.h.resize(total / ch);
    for (auto& x : s.h) CK(hipMemCreate(&x, ch, &prop, 0));
    return s;
}
void map_at(const Slab& s, char* va) {
    for (size_t i = 0; i < s.h.size(); ++i) CK(hipMemMap(va + s.ch * i, s.ch, 0, s.h[i], 0));
    CK(hipMemSetAccess(va, s.bytes(), &acc, 1));
}
void unmap_at(const Slab& s, char* va) { for (size_t i = 0; i < s.h.size(); ++i) CK(hipMemUnmap(va + s.ch * i, s.ch)); }

int main() {
    CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, GB = 1ull << 30;
    int dev = 0; CK(hipGetDevice(&dev));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const int NB = 48;
    char* R = nullptr;
    CK(hipMemAddressReserve((void**)&R, (NB + 2) * GB, 1 * GB, nullptr, 0));
    printf("reserved %d GB at %p (1 GB aligned: %s)\n", NB + 2, (void*)R, ((uintptr_t)R & (GB - 1)) ? "NO" : "yes");
    Slab one = make(1 * GB, 128 * MB);
    std::vector<double> m(NB);
    for (int pass = 0; pass < 2; ++pass) {
        printf("pass %d: write GB/s of the SAME 1 GB of physical memory mapped at virtual block k:\n", pass);
        for (int k = 0; k < NB; ++k) {
            map_at(one, R + k * GB);
            m[k] = wr_bw(R + k * GB, GB);
            unmap_at(one, R + k * GB);
            printf(" %.0f", m[k]);
            if (k % 16 == 15) printf("\n");
        }
        fflush(stdout);
    }
    int kf = (int)(std::max_element(m.begin(), m.end()) - m.begin()), ks = (int)(std::min_element(m.begin(), m.end()) - m.begin());
    printf("fastest block %d (%.0f), slowest block %d (%.0f)\n", kf, m[kf], ks, m[ks]);
    for (int k : {kf, ks}) {
        map_at(one, R + k * GB);
        printf("3. block %d: interleaved sweep %.0f, XCD-partitioned sweep %.0f GB/s\n", k, wr_bw(R + k * GB, GB), wr_bw(R + k * GB, GB, true));
        unmap_at(one, R + k * GB);
    }
    Slab other = make(1 * GB, 128 * MB);
    Slab big = make(1 * GB, 1 * GB);
    for (int k : {kf, ks}) {
        map_at(other, R + k * GB);
        printf("4. block %d with OTHER physical memory (8 x 128 MB): %.0f", k, wr_bw(R + k * GB, GB));
        unmap_at(other, R + k * GB);
        map_at(big, R + k * GB);
        printf(";  one 1 GB handle: %.0f GB/s\n", wr_bw(R + k * GB, GB));
        unmap_at(big, R + k * GB);
    }
    // 5. windows sliding across the border next to the fastest block
    if (kf + 2 < NB && kf > 0) {
        Slab three = make(3 * GB, 128 * MB);
        map_at(three, R + (kf - 1) * GB);
        printf("5. 1 GB windows from block %d - 1, step 256 MB:", kf);
        for (int q = 0; q <= 8; ++q) printf(" %.0f", wr_bw(R + (kf - 1) * GB + q * 256 * MB, GB));
        printf("\n");
        unmap_at(three, R + (kf - 1) * GB);
        for (auto& x : three.h) CK(hipMemRelease(x));
    }
    // 6. free the reservation, reserve again
    CK(hipMemAddressFree(R, (NB + 2) * GB));
    char* R2 = nullptr;
    CK(hipMemAddressReserve((void**)&R2, (NB + 2) * GB, 1 * GB, nullptr, 0));
    printf("6. second reservation at %p (%s):\n", (void*)R2, R2 == R ? "same address" : "different address");
    for (int k = 0; k < NB; ++k) {
        map_at(one, R2 + k * GB);
        printf(" %.0f", wr_bw(R2 + k * GB, GB));
        unmap_at(one, R2 + k * GB);
        if (k % 16 == 15) printf("\n");
    }
    // 7. a reservation at a far-away hint address
    char* R3 = nullptr;
    if (hipMemAddressReserve((void**)&R3, 18 * GB, 1 * GB, (void*)0x100000000000ull, 0) == hipSuccess) {
        printf("7. third reservation (hint 0x100000000000) at %p:\n", (void*)R3);
        for (int k = 0; k < 16; ++k) {
            map_at(one, R3 + k * GB);
            printf(" %.0f", wr_bw(R3 + k * GB, GB));
            unmap_at(one, R3 + k * GB);
        }
        printf("\n");
    }
    return 0;
}
