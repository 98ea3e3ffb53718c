// assign_owner_probe.hip — which form of the last-writer ("owner") pass of dxo_assign is fastest? (csrc/assign.hip: assign_owner, 0.51 ms of the
// direct form's 0.65 ms on the bench leg's case: Q2 hexahedra 108^3, 3.4e7 entries -> 10 218 313 dofs.)
//   A  shipped: 64-bit atomicMax(owner[d], e + 1), ascending e
//   B  32-bit words (entries < 2^32)
//   C  32-bit, DEScending e, and a plain load first: an entry that already sees a larger owner skips its atomic (a stale load can only
//      under-read a monotone word, so skipping is safe)
//   D  32-bit, ascending e, same filter (control: later entries are larger, the filter should rarely fire)
//   E  no atomics: racing plain stores, then "store again if the word is smaller than mine" rounds until nothing changes
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp/assign_owner_probe.hip -o scripts/exp/assign_owner_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void owner64(const int32_t* __restrict__ dofs, unsigned long long* __restrict__ owner, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) atomicMax(owner + dofs[e], (unsigned long long)(e + 1));
}
__global__ __launch_bounds__(256) void owner32(const int32_t* __restrict__ dofs, uint32_t* __restrict__ owner, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) atomicMax(owner + dofs[e], (uint32_t)(e + 1));
}
template <bool REV>
__global__ __launch_bounds__(256) void owner32_filter(const int32_t* __restrict__ dofs, uint32_t* __restrict__ owner, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t e = REV ? n - 1 - i : i;
        uint32_t* w = owner + dofs[e];
        if (__builtin_nontemporal_load(w) < (uint32_t)(e + 1)) atomicMax(w, (uint32_t)(e + 1));
    }
}
__global__ __launch_bounds__(256) void owner32_plain(const int32_t* __restrict__ dofs, uint32_t* __restrict__ owner, int64_t n, int first, unsigned* __restrict__ changed) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool any = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        uint32_t* w = owner + dofs[e];
        if (first || *w < (uint32_t)(e + 1)) { *w = (uint32_t)(e + 1); any = true; }
    }
    if (!first && any) *changed = 1u;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 108, N = 2 * n + 1;
    const int64_t nc = (int64_t)n * n * n, nd = (int64_t)N * N * N, ne = nc * 27;
    std::vector<int32_t> dofs((size_t)ne);
    for (int64_t c = 0; c < nc; ++c) {
        const int i = (int)(c % n), j = (int)((c / n) % n), k = (int)(c / ((int64_t)n * n));
        for (int a = 0; a < 27; ++a) dofs[(size_t)(c * 27 + a)] = (int32_t)((((int64_t)(2 * k + a / 9)) * N + 2 * j + (a / 3) % 3) * N + 2 * i + a % 3);
    }
    int32_t* d_dofs; unsigned long long* o64; uint32_t *o32, *o32b; unsigned* d_changed;
    CK(hipMalloc(&d_dofs, ne * 4)); CK(hipMalloc(&o64, nd * 8)); CK(hipMalloc(&o32, nd * 4)); CK(hipMalloc(&o32b, nd * 4)); CK(hipMalloc(&d_changed, 4));
    CK(hipMemcpy(d_dofs, dofs.data(), ne * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, nullptr);
            for (int l = 0; l < 10; ++l) launch();
            (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms / 10);
        }
        return best;
    };
    std::vector<uint32_t> ref((size_t)nd), got((size_t)nd);
    std::vector<unsigned long long> r64((size_t)nd);
    const float tA = time([&] { (void)hipMemsetAsync(o64, 0, nd * 8, nullptr); hipLaunchKernelGGL(owner64, dim3(grid), dim3(256), 0, nullptr, d_dofs, o64, ne); });
    CK(hipMemcpy(r64.data(), o64, nd * 8, hipMemcpyDeviceToHost));
    for (int64_t d = 0; d < nd; ++d) ref[(size_t)d] = (uint32_t)r64[(size_t)d];
    auto same = [&](uint32_t* dev) { (void)hipMemcpy(got.data(), dev, nd * 4, hipMemcpyDeviceToHost); return got == ref; };
    const float tB = time([&] { (void)hipMemsetAsync(o32, 0, nd * 4, nullptr); hipLaunchKernelGGL(owner32, dim3(grid), dim3(256), 0, nullptr, d_dofs, o32, ne); });
    const bool okB = same(o32);
    const float tC = time([&] { (void)hipMemsetAsync(o32, 0, nd * 4, nullptr); hipLaunchKernelGGL(owner32_filter<true>, dim3(grid), dim3(256), 0, nullptr, d_dofs, o32, ne); });
    const bool okC = same(o32);
    const float tD = time([&] { (void)hipMemsetAsync(o32, 0, nd * 4, nullptr); hipLaunchKernelGGL(owner32_filter<false>, dim3(grid), dim3(256), 0, nullptr, d_dofs, o32, ne); });
    const bool okD = same(o32);
    // E: rounds until the fixed point (host reads the flag each round, as the product would have to)
    int rounds = 0;
    auto runE = [&] {
        (void)hipMemsetAsync(o32b, 0, nd * 4, nullptr);
        hipLaunchKernelGGL(owner32_plain, dim3(grid), dim3(256), 0, nullptr, d_dofs, o32b, ne, 1, d_changed);
        rounds = 1;
        for (;;) {
            (void)hipMemsetAsync(d_changed, 0, 4, nullptr);
            hipLaunchKernelGGL(owner32_plain, dim3(grid), dim3(256), 0, nullptr, d_dofs, o32b, ne, 0, d_changed);
            unsigned ch = 0; (void)hipMemcpy(&ch, d_changed, 4, hipMemcpyDeviceToHost);
            ++rounds;
            if (!ch || rounds > 40) break;
        }
    };
    const float tE = time(runE);
    const bool okE = same(o32b);
    std::printf("{\"entries\": %lld, \"dofs\": %lld, \"A_atomic64_ms\": %.4f, \"B_atomic32_ms\": %.4f, \"B_same\": %s, \"C_atomic32_descending_filter_ms\": %.4f, \"C_same\": %s, "
                "\"D_atomic32_ascending_filter_ms\": %.4f, \"D_same\": %s, \"E_plain_rounds_ms\": %.4f, \"E_rounds\": %d, \"E_same\": %s}\n",
                (long long)ne, (long long)nd, tA, tB, okB ? "true" : "false", tC, okC ? "true" : "false", tD, okD ? "true" : "false", tE, rounds, okE ? "true" : "false");
    return 0;
}
