// place_exp11.hip — is the fast class a matter of physical SCATTER? (place_exp10: physically contiguous blocks are
// uniformly slow, ordinary allocations are fast about one time in six.) Slabs of 3.44 GB built with the virtual-memory
// API from 2 MB physical chunks, each slab mapped ONCE at its own fresh virtual range (no remapping):
//   seq    chunks mapped in creation order
//   perm   the same number of chunks mapped in a random permutation
//   gap    every second chunk of a pool twice the size (the others stay unmapped): physically gapped
//   big    128 MB chunks in creation order
//   plain  hipMalloc
// REP slabs of each kind, all alive at once; pure stores and the 13:43 mix as in place_exp10.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <chrono>
#include <random>
#include <string>
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
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};
const size_t MB = 1ull << 20;
struct Slab { char* va = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> h; const char* kind = ""; };

Slab make(const char* kind, size_t bytes, size_t chunk, int mode, std::mt19937& rng, char* va = nullptr) {
    Slab s; s.kind = kind;
    const size_t nch = (bytes + chunk - 1) / chunk;
    s.bytes = nch * chunk;
    s.va = va;
    const size_t stride = mode >= 2 ? (size_t)mode : 1;
    const size_t pool = stride * nch;
    s.h.resize(pool);
    for (auto& x : s.h) CK(hipMemCreate(&x, chunk, &prop, 0));
    std::vector<size_t> order(nch);
    for (size_t i = 0; i < nch; ++i) order[i] = stride * i;
    if (mode == 1) std::shuffle(order.begin(), order.end(), rng);
    if (!s.va) CK(hipMemAddressReserve((void**)&s.va, s.bytes, 2 * MB, nullptr, 0));
    for (size_t i = 0; i < nch; ++i) CK(hipMemMap(s.va + i * chunk, chunk, 0, s.h[order[i]], 0));
    CK(hipMemSetAccess(s.va, s.bytes, &acc, 1));
    if (stride > 1) {   // give the unmapped chunks back: the mapped ones keep their physical places
        std::vector<hipMemGenericAllocationHandle_t> keep;
        for (size_t i = 0; i < pool; ++i) {
            if (i % stride == 0) keep.push_back(s.h[i]);
            else CK(hipMemRelease(s.h[i]));
        }
        s.h.swap(keep);
    }
    return s;
}
int main(int argc, char** argv) {
    const int REP = argc > 1 ? atoi(argv[1]) : 4;
    const char* order = argc > 2 ? argv[2] : "spgbP";   // s seq, p perm, g gap, b big, P plain hipMalloc: creation order inside a round
    const int va_first = argc > 3 ? atoi(argv[3]) : 0;   // 1: reserve every virtual range up front, then create + map in REVERSE order
    const long tiles = 10000000 / 64;
    const size_t out_bytes = (size_t)tiles * 43 * 1024, in_bytes = (size_t)tiles * 13 * 1024;
    CK(hipStreamCreate(&st));
    int dev = 0; CK(hipGetDevice(&dev));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* src = nullptr;
    CK(hipMalloc(&src, in_bytes));
    CK(hipMemset(src, 0, in_bytes));
    std::mt19937 rng(7);
    std::vector<Slab> slabs;
    // order = comma-separated kinds: "<chunk MB>:<stride>" (stride 1 = consecutive chunks) or "P" (hipMalloc)
    std::vector<std::string> kinds;
    {
        std::string o = order, cur;
        for (char ch : o + ",") {
            if (ch == ',') { if (!cur.empty()) kinds.push_back(cur); cur.clear(); }
            else cur.push_back(ch);
        }
    }
    std::vector<std::string> names;
    for (int r = 0; r < REP; ++r)
        for (const auto& k : kinds) {
            const auto t0 = std::chrono::steady_clock::now();
            if (k == "P") {
                Slab p; p.kind = "plain"; p.bytes = out_bytes; CK(hipMalloc((void**)&p.va, out_bytes)); slabs.push_back(p);
            } else {
                const size_t chunk_mb = std::stoul(k.substr(0, k.find(':')));
                const int stride = std::stoi(k.substr(k.find(':') + 1));
                slabs.push_back(make("", out_bytes, chunk_mb * MB, stride >= 2 ? stride : 0, rng));
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            names.push_back(k + " (" + std::to_string((int)ms) + " ms to build)");
        }
    (void)va_first;
    for (size_t i = 0; i < slabs.size(); ++i) slabs[i].kind = names[i].c_str();
    for (int pass = 0; pass < 1; ++pass)
        for (const auto& s : slabs) {
            float w = timeit(4, 3, [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, st, tiles, (f64x2*)s.va); });
            float m = timeit(4, 3, [&] { hipLaunchKernelGGL(mix, dim3(4096), dim3(256), 0, st, tiles, (const f64x2*)src, (f64x2*)s.va); });
            printf("pass %d  %-28s va %p  write %5.0f  mix %5.0f GB/s\n", pass, s.kind, (void*)s.va, out_bytes / w / 1e6, (out_bytes + in_bytes) / m / 1e6);
            fflush(stdout);
        }
    CK(hipDeviceSynchronize());
    return 0;   // process exit releases the mappings
}
