// assign_sorted_probe.hip — VERDICT r05 next #7 as a measurement: does a plan SORTED BY SOURCE POSITION make dxo_assign_apply faster?
// Case of the bench leg assign_cg: Q2 hexahedra 108^3, values[cell][27] (3.4e7 doubles, 272 MB) -> 10 218 313 dofs, NumPy's last-writer rule.
//   A  the shipped form:   for d in dof order:            coeff[d]        = values[src[d]]        (4-byte plan entry, coalesced stores, gathered loads)
//   B  sorted by source:   for i in source order:         coeff[dst_s[i]] = values[src_s[i]]      (8-byte plan entry, near-coalesced loads, scattered stores)
// Build (CPU container): hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp/assign_sorted_probe.hip -o scripts/exp/assign_sorted_probe
// Run on the GPU box: scripts/exp/assign_sorted_probe [n]     (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around it gives the traffic)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void apply_dof_order(const int32_t* __restrict__ src, const double* __restrict__ values, double* __restrict__ coeff, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += stride) {
        const int32_t s = src[d];
        if (s >= 0) coeff[d] = values[s];
    }
}

__global__ __launch_bounds__(256) void apply_source_order(const int32_t* __restrict__ src_s, const int32_t* __restrict__ dst_s, const double* __restrict__ values,
                                                          double* __restrict__ coeff, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) coeff[dst_s[i]] = values[src_s[i]];
}

// B4: four consecutive pairs per thread (16-byte index loads, four gathers in flight)
__global__ __launch_bounds__(256) void apply_source_order_x4(const int4* __restrict__ src_s, const int4* __restrict__ dst_s, const double* __restrict__ values,
                                                             double* __restrict__ coeff, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const int4 s = src_s[i], d = dst_s[i];
        const double v0 = values[s.x], v1 = values[s.y], v2 = values[s.z], v3 = values[s.w];
        coeff[d.x] = v0; coeff[d.y] = v1; coeff[d.z] = v2; coeff[d.w] = v3;
    }
}

int main(int argc, char** argv) {
    // argv[2] = 1: the numbering of the bench leg's mesh (tools/synthetic.py structured_mesh: cells and lattice nodes with z fastest, local nodes
    // with x fastest) instead of x fastest everywhere — neighbouring dofs then come from local nodes 9 apart, not 1 apart
    const int n = argc > 1 ? std::atoi(argv[1]) : 108, N = 2 * n + 1;
    const bool zfast = argc > 2 && std::atoi(argv[2]) == 1;
    const int64_t nc = (int64_t)n * n * n, nd = (int64_t)N * N * N;
    std::vector<int32_t> src((size_t)nd, -1);
    for (int64_t c = 0; c < nc; ++c) {      // ascending cell: the last writer wins
        int i = (int)(c % n), j = (int)((c / n) % n), k = (int)(c / ((int64_t)n * n));
        if (zfast) std::swap(i, k);
        for (int a = 0; a < 27; ++a) {
            const int I = 2 * i + a % 3, J = 2 * j + (a / 3) % 3, K = 2 * k + a / 9;
            const int64_t dof = zfast ? ((int64_t)I * N + J) * N + K : ((int64_t)K * N + J) * N + I;
            src[(size_t)dof] = (int32_t)(c * 27 + a);
        }
    }
    std::vector<int32_t> order((size_t)nd);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return src[(size_t)a] < src[(size_t)b]; });
    std::vector<int32_t> src_s((size_t)nd), dst_s((size_t)nd);
    for (int64_t i = 0; i < nd; ++i) { dst_s[(size_t)i] = order[(size_t)i]; src_s[(size_t)i] = src[(size_t)order[(size_t)i]]; }
    int32_t *d_src, *d_src_s, *d_dst_s;
    double *d_val, *d_c1, *d_c2;
    CK(hipMalloc(&d_src, nd * 4)); CK(hipMalloc(&d_src_s, nd * 4)); CK(hipMalloc(&d_dst_s, nd * 4));
    CK(hipMalloc(&d_val, nc * 27 * 8)); CK(hipMalloc(&d_c1, nd * 8)); CK(hipMalloc(&d_c2, nd * 8));
    CK(hipMemcpy(d_src, src.data(), nd * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_src_s, src_s.data(), nd * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dst_s, dst_s.data(), nd * 4, hipMemcpyHostToDevice));
    std::vector<double> val((size_t)(nc * 27));
    for (size_t e = 0; e < val.size(); ++e) val[e] = (double)(e % 1000003) * 1e-3;
    CK(hipMemcpy(d_val, val.data(), val.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_c1, 0, nd * 8)); CK(hipMemset(d_c2, 0, nd * 8));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0, nullptr);
            for (int l = 0; l < 20; ++l) launch();
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms / 20);
        }
        return best;
    };
    const float tA = time([&] { hipLaunchKernelGGL(apply_dof_order, dim3(grid), dim3(256), 0, nullptr, d_src, d_val, d_c1, nd); });
    const float tB = time([&] { hipLaunchKernelGGL(apply_source_order, dim3(grid), dim3(256), 0, nullptr, d_src_s, d_dst_s, d_val, d_c2, nd); });
    const float tB2 = time([&] { hipLaunchKernelGGL(apply_source_order, dim3(grid * 2), dim3(256), 0, nullptr, d_src_s, d_dst_s, d_val, d_c2, nd); });
    const float tB8 = time([&] { hipLaunchKernelGGL(apply_source_order, dim3(grid * 8), dim3(256), 0, nullptr, d_src_s, d_dst_s, d_val, d_c2, nd); });
    const float tB4 = time([&] { hipLaunchKernelGGL(apply_source_order_x4, dim3(grid), dim3(256), 0, nullptr, (const int4*)d_src_s, (const int4*)d_dst_s, d_val, d_c2, nd / 4); });
    const float tA8 = time([&] { hipLaunchKernelGGL(apply_dof_order, dim3(grid * 8), dim3(256), 0, nullptr, d_src, d_val, d_c1, nd); });
    std::printf("{\"source_order_grid_x2_ms\": %.4f, \"source_order_grid_x8_ms\": %.4f, \"source_order_4_per_thread_ms\": %.4f, \"dof_order_grid_x8_ms\": %.4f}\n", tB2, tB8, tB4, tA8);
    std::vector<double> c1((size_t)nd), c2((size_t)nd);
    CK(hipMemcpy(c1.data(), d_c1, nd * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c2.data(), d_c2, nd * 8, hipMemcpyDeviceToHost));
    const bool same = c1 == c2;
    const double alg = (double)nd * 20.0;      // 4-byte source position + the winning value + the coefficient entry per dof
    std::printf("{\"numbering\": \"%s\", \"cells\": %lld, \"dofs\": %lld, \"dof_order_ms\": %.4f, \"source_order_ms\": %.4f, \"identical\": %s, \"algorithmic_MB\": %.1f, "
                "\"dof_order_TBps_on_algorithmic\": %.2f, \"source_order_TBps_on_algorithmic\": %.2f}\n",
                zfast ? "z fastest (the bench leg's mesh)" : "x fastest", (long long)nc, (long long)nd, tA, tB, same ? "true" : "false", alg / 1e6, alg / tA / 1e9, alg / tB / 1e9);
    return 0;
}
