// Host half of the DXO_MEM_HOST pipeline on its own: vm_host_rebuild_range<6> over n points with the context's worker
// pool, threads x grain sweep, on (a) malloc'd memory first-touched by the workers, (b) hipHostMalloc memory.
// build: hipcc -O3 -std=c++17 -Idolfinx_external_operator_amd/csrc scripts/exp/archive/host_rebuild_bench.hip -o scripts/exp/host_rebuild_bench -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "host_pool.h"
#include "vm_host.h"

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int D = 6;
    const VmHostConst hc{40384.6, 53846.2, 80769.2, 0.99};
    for (int pinned = 0; pinned < 2; ++pinned) {
        double *sigma, *dp, *C;
        if (pinned) {
            if (hipHostMalloc((void**)&sigma, n * D * 8, 0) != hipSuccess || hipHostMalloc((void**)&dp, n * 8, 0) != hipSuccess ||
                hipHostMalloc((void**)&C, n * D * D * 8, 0) != hipSuccess) {
                printf("hipHostMalloc failed\n");
                return 1;
            }
        } else {
            sigma = (double*)aligned_alloc(64, n * D * 8);
            dp = (double*)aligned_alloc(64, n * 8);
            C = (double*)aligned_alloc(64, n * D * D * 8);
        }
        dxo_host_pool* pool = nullptr;
        // parallel first touch / fill
        dxo_pool_parallel_for(pool, 64, n, 65536, [&](int64_t b, int64_t e) {
            std::mt19937_64 g(b);
            std::normal_distribution<double> N(0.0, 100.0);
            for (int64_t i = b; i < e; ++i) {
                for (int k = 0; k < D; ++k) sigma[i * D + k] = N(g);
                dp[i] = (i & 3) ? 1e-3 : 0.0;
            }
            memset(C + b * D * D, 0, (e - b) * D * D * 8);
        });
        for (int threads : {16, 32, 64}) {
            for (int64_t grain : {512, 2048}) {
                for (int64_t chunk : {(int64_t)65536, (int64_t)131072, n}) {   // the pipeline hands over chunks of 2^16 points
                    double best = 1e30;
                    for (int rep = 0; rep < 3; ++rep) {
                        const auto t0 = std::chrono::steady_clock::now();
                        for (int64_t c0 = 0; c0 < n; c0 += chunk) {
                            const int64_t m = n - c0 < chunk ? n - c0 : chunk;
                            dxo_pool_parallel_for(pool, threads, m, grain, [&](int64_t b, int64_t e) {
                                vm_host_rebuild_range<6>(hc, sigma + c0 * D, dp + c0, C + c0 * D * D, b, e);
                            });
                        }
                        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                        if (ms < best) best = ms;
                    }
                    printf("%s threads %3d grain %5ld chunk %8ld: %7.2f ms  %6.1f GB/s (344 B/pt)  %.2e pt/s\n", pinned ? "pinned" : "malloc",
                           threads, (long)grain, (long)chunk, best, 344.0 * n / best / 1e6, n / best * 1e3);
                    fflush(stdout);
                }
            }
        }
        dxo_host_pool_destroy(pool);
        if (pinned) { (void)hipHostFree(sigma); (void)hipHostFree(dp); (void)hipHostFree(C); }
        else { free(sigma); free(dp); free(C); }
    }
    return 0;
}
