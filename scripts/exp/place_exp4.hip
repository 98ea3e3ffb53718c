// place_exp4.hip — place_exp3 showed: the same physical chunks are slow in one mapping and fast in another, so the
// write speed of a slab is a property of its MAPPING (virtual address / page-table shape), not of the physical memory.
// Which property?  One set of physical memory, many mappings:
//   a. 24 chunks of 128 MB mapped at a 4 GB-aligned virtual address R                         -> write / mix bandwidth
//   b. the same chunks remapped at R + k * 128 MB (k = 1..8), at R + 2 MB, R + 64 MB          -> VA alignment
//   c. a second mapping while the first one exists                                            -> "second mapping" effect
//   d. chunk size 2 MB ... 3 GB (one handle)                                                   -> page-table fragment size
//   e. four hipMalloc(3 GB) in the same process with their virtual addresses                   -> the plain allocation
//   f. 360 more chunks created and mapped behind the slab, then (a) again                      -> neighbours / table placement
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
static char* g_src;
double wr_bw(void* p, size_t bytes) {
    const long n = bytes / 16384;
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, n, (f64x2*)p); });
    return bytes / ms / 1e6;
}
double mix_bw(void* dst, size_t dst_bytes) {
    const long n_tiles = dst_bytes / (43 * 1024);
    float ms = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(4096), dim3(256), 0, st, n_tiles, (const f64x2*)g_src, (f64x2*)dst); });
    return n_tiles * 56.0 * 1024 / ms / 1e6;
}

static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
struct Slab {
    std::vector<hipMemGenericAllocationHandle_t> h;
    size_t ch = 0;
    size_t bytes() const { return ch * h.size(); }
};
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
void drop(Slab& s) { for (auto& x : s.h) CK(hipMemRelease(x)); s.h.clear(); }
void report(const char* what, char* va, size_t bytes) {
    printf("%-58s va %p (mod 4 GB: %7.1f MB)  write %5.0f  mix %5.0f GB/s\n", what, (void*)va, ((uintptr_t)va & 0xffffffffull) / 1048576.0, wr_bw(va, bytes), mix_bw(va, bytes));
    fflush(stdout);
}

int main() {
    CK(hipStreamCreate(&st));
    const size_t MB = 1ull << 20, GB = 1ull << 30, B3 = 3 * GB;
    int dev = 0; CK(hipGetDevice(&dev));
    CK(hipMalloc(&g_src, 1 * GB));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    char* R = nullptr;
    CK(hipMemAddressReserve((void**)&R, 80 * GB, 4 * GB, nullptr, 0));
    printf("reserved 80 GB at %p\n", (void*)R);

    // e. plain hipMalloc first (what round 1 measured)
    {
        std::vector<char*> b(4);
        for (auto& x : b) CK(hipMalloc(&x, B3));
        for (auto& x : b) report("e. hipMalloc(3 GB)", x, B3);
        for (auto& x : b) CK(hipFree(x));
    }
    // a, b, c
    Slab s = make(B3, 128 * MB);
    map_at(s, R);
    report("a. 24 x 128 MB chunks at R", R, B3);
    report("a. again", R, B3);
    unmap_at(s, R);
    for (size_t off : {128 * MB, 256 * MB, 384 * MB, 512 * MB, 640 * MB, 768 * MB, 896 * MB, 1024 * MB, 2 * MB, 4 * MB, 64 * MB, 2048 * MB + 2 * MB}) {
        map_at(s, R + off);
        char name[96]; snprintf(name, sizeof name, "b. same chunks at R + %zu MB", off / MB);
        report(name, R + off, B3);
        unmap_at(s, R + off);
    }
    map_at(s, R);
    map_at(s, R + 8 * GB);
    report("c. first mapping (R) while a second exists", R, B3);
    report("c. second mapping (R + 8 GB)", R + 8 * GB, B3);
    unmap_at(s, R + 8 * GB);
    unmap_at(s, R);
    {   // reversed chunk order at R
        Slab r = s; std::reverse(r.h.begin(), r.h.end());
        map_at(r, R);
        report("b. chunks in REVERSED order at R", R, B3);
        unmap_at(r, R);
    }
    drop(s);
    // d. chunk size
    for (size_t ch : {2 * MB, 8 * MB, 32 * MB, 128 * MB, 512 * MB, 1024 * MB, 3072 * MB}) {
        Slab t = make(B3, ch);
        map_at(t, R);
        char name[96]; snprintf(name, sizeof name, "d. fresh chunks of %zu MB at R", ch / MB);
        report(name, R, B3);
        unmap_at(t, R);
        map_at(t, R + 2 * MB);
        snprintf(name, sizeof name, "d. the same at R + 2 MB");
        report(name, R + 2 * MB, B3);
        unmap_at(t, R + 2 * MB);
        drop(t);
    }
    // f. with many neighbours
    Slab a = make(B3, 128 * MB);
    map_at(a, R);
    report("f. slab at R before the neighbours", R, B3);
    Slab nb = make(45 * GB, 128 * MB);
    map_at(nb, R + 4 * GB);
    report("f. slab at R with 45 GB mapped at R + 4 GB", R, B3);
    for (int k = 0; k < 12; ++k) {
        char name[96]; snprintf(name, sizeof name, "f. neighbour window at R + 4 GB + %d x 1 GB + 640 MB", k);
        report(name, R + 4 * GB + k * GB + 640 * MB, 1 * GB);
    }
    for (int k = 0; k < 6; ++k) {
        char name[96]; snprintf(name, sizeof name, "f. neighbour 3 GB window at R + 4 GB + %d x 3 GB", k);
        report(name, R + 4 * GB + k * 3 * GB, 3 * GB);
    }
    return 0;
}
