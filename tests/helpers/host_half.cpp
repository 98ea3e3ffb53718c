// Host half of the DXO_MEM_HOST pipeline, built WITHOUT HIP for the CPU test-suite:
//   * as a shared library (host_rebuild): csrc/vm_host.h on the worker pool of csrc/host_pool.h, so the product's
//     host-side tangent rebuild can be compared with the oracle on a box without a GPU;
//   * as a program (-DHOST_HALF_MAIN) under -fsanitize=thread / address,undefined: many parallel_for rounds of varying
//     size and thread count (pool re-creation included), sums checked.
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "host_pool.h"
#include "vm_host.h"

// form: 0 = the product's run-time choice, 1 = portable SSE2 form, 2 = AVX2 + FMA form (-3 if the CPU lacks it)
extern "C" int host_rebuild_form(int form, int d, int64_t n, double E, double nu, double H, int threads, const double* sigma,
                                 double* dp, double* C_tang) {
    const double lmbda = E * nu / (1.0 + nu) / (1.0 - 2.0 * nu), mu = E / 2.0 / (1.0 + nu);
    const VmHostConst c{lmbda, 2.0 * mu, 3 * mu, 3 * mu / (3 * mu + H)};
    if (d != 4 && d != 6) return -2;
    if (form == 2 && !(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"))) return -3;
    dxo_host_pool* pool = nullptr;
    dxo_pool_parallel_for(pool, threads, n, 256, [&](int64_t b, int64_t e) {
        if (form == 1) {
            if (d == 4) vm_host_rebuild_range_sse2<4>(c, sigma, dp, C_tang, b, e);
            else vm_host_rebuild_range_sse2<6>(c, sigma, dp, C_tang, b, e);
        } else if (form == 2) {
            if (d == 4) vm_host_rebuild_range_avx2<4>(c, sigma, dp, C_tang, b, e);
            else vm_host_rebuild_range_avx2<6>(c, sigma, dp, C_tang, b, e);
        } else {
            if (d == 4) vm_host_rebuild_range<4>(c, sigma, dp, C_tang, b, e);
            else vm_host_rebuild_range<6>(c, sigma, dp, C_tang, b, e);
        }
    });
    dxo_host_pool_destroy(pool);
    return 0;
}

extern "C" int host_rebuild(int d, int64_t n, double E, double nu, double H, int threads, const double* sigma, double* dp,
                            double* C_tang) {
    const double lmbda = E * nu / (1.0 + nu) / (1.0 - 2.0 * nu), mu = E / 2.0 / (1.0 + nu);   // demo_plasticity_von_mises.py:190-191
    const VmHostConst c{lmbda, 2.0 * mu, 3 * mu, 3 * mu / (3 * mu + H)};
    if (d != 4 && d != 6) return -2;
    dxo_host_pool* pool = nullptr;
    dxo_pool_parallel_for(pool, threads, n, 256, [&](int64_t b, int64_t e) {
        if (d == 4) vm_host_rebuild_range<4>(c, sigma, dp, C_tang, b, e);
        else vm_host_rebuild_range<6>(c, sigma, dp, C_tang, b, e);
    });
    dxo_host_pool_destroy(pool);
    return 0;
}

#ifdef HOST_HALF_MAIN
int main() {
    dxo_host_pool* pool = nullptr;
    std::vector<double> out(100000);
    long rounds = 0;
    for (int threads : {1, 2, 5, 3, 8, 8, 2}) {
        for (int64_t n : {int64_t(0), int64_t(1), int64_t(63), int64_t(4097), int64_t(100000)}) {
            for (int rep = 0; rep < 6; ++rep) {
                std::fill(out.begin(), out.end(), 0.0);
                dxo_pool_parallel_for(pool, threads, n, 64 + 37 * rep, [&](int64_t b, int64_t e) {
                    for (int64_t i = b; i < e; ++i) out[(size_t)i] += (double)(i + 1);
                });
                const double want = 0.5 * (double)n * (double)(n + 1);
                const double got = std::accumulate(out.begin(), out.begin() + n, 0.0);
                if (got != want) { std::printf("round %ld: threads %d n %ld: sum %.17g != %.17g\n", rounds, threads, (long)n, got, want); return 1; }
                ++rounds;
            }
        }
    }
    dxo_host_pool_destroy(pool);
    // the rebuild itself on the pool: every entry written exactly once (guard words stay)
    const int64_t n = 5001;
    std::vector<double> sigma((size_t)n * 6), dp((size_t)n), C((size_t)n * 36 + 8, -7.0);
    unsigned s = 1u;
    for (auto& v : sigma) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 16777216.0 - 0.5) * 600.0; }
    for (int64_t i = 0; i < n; ++i) dp[(size_t)i] = (i % 3) ? 1e-4 * (double)(i % 7) : 0.0;
    dp[17] = -0.0;   // the kernel's mark for the reference's 0/0 point
    if (host_rebuild(6, n, 70e3, 0.3, 707.07, 4, sigma.data(), dp.data(), C.data())) return 2;
    for (size_t k = (size_t)n * 36; k < C.size(); ++k) if (C[k] != -7.0) return 3;
    if (!(C[17 * 36] != C[17 * 36]) || std::signbit(dp[17])) return 4;   // NaN tangent, dp back to +0
    for (int64_t i = 0; i < n * 36; ++i) if (i / 36 != 17 && !(C[(size_t)i] == C[(size_t)i])) return 5;
    // the CPU budget the pool caps itself with: at least 1, at most the machine's threads, and overridable
    const int budget = dxo_host_cpu_budget();
    if (budget < 1 || (std::thread::hardware_concurrency() > 0 && budget > (int)std::thread::hardware_concurrency())) return 6;
    setenv("DXO_HOST_CPU_BUDGET", "3", 1);
    if (dxo_host_cpu_budget() != 3) return 7;
    unsetenv("DXO_HOST_CPU_BUDGET");
    std::printf("host half harness: ok (%ld parallel_for rounds, cpu budget %d)\n", rounds, budget);
    return 0;
}
#endif
