// vm_host.h — the host half of DXO_MEM_HOST dxo_von_mises with option "vm_host_tangent" = 1: consistent tangent from
// the RETURNED state (sigma, dp) on the CPU (plain C++17 + SSE2, no HIP types: also built on its own for the CPU tests
// and the sanitizers). Same formulas and operation order as vm_tangent_state / vm_store_tangent (von_mises.hip),
// reference: demo_plasticity_von_mises.py:318-324 with s = dev sigma.
#pragma once

#include <emmintrin.h>

#include <cmath>
#include <cstdint>
#include <limits>

struct VmHostConst {
    double lmbda, mu2, mu3;   // lambda, 2 mu, 3 mu
    double ratio;             // 3 mu / (3 mu + H)
};

template <int D>
void vm_host_rebuild_range(const VmHostConst& c, const double* __restrict__ sigma, double* __restrict__ dp,
                           double* __restrict__ C_tang, int64_t b, int64_t e) {
    const bool stream = (((uintptr_t)C_tang) & 15u) == 0;   // D*D*8 is a multiple of 16: every point block is aligned
    for (int64_t i = b; i < e; ++i) {
        const double* sg = sigma + i * D;
        const double mean = (sg[0] + sg[1] + sg[2]) * (1.0 / 3.0);
        double s[D], nrm[D];
        for (int k = 0; k < D; ++k) s[k] = k < 3 ? sg[k] - mean : sg[k];
        double ss = 0.0;
        for (int k = 0; k < D; ++k) ss += s[k] * s[k];
        const double sigma_eq = std::sqrt(3.0 / 2.0 * ss);
        double dpi = dp[i];
        if (dpi == 0.0 && std::signbit(dpi)) {
            // the kernel's mark for f_elastic == 0 exactly: the reference's n_elas = s/sigma_eq * 0/0 is NaN there (:318)
            // and with it every tangent entry; dp itself is +0 in the reference
            dp[i] = 0.0;
            double* Cn = C_tang + i * (D * D);
            for (int k = 0; k < D * D; ++k) Cn[k] = std::numeric_limits<double>::quiet_NaN();
            continue;
        }
        const double beta = c.mu3 * dpi / (sigma_eq + c.mu3 * dpi);
        const double ind = dpi > 0.0 ? 1.0 : 0.0;
        for (int k = 0; k < D; ++k) nrm[k] = s[k] / sigma_eq * ind;
        const double a = c.mu3 * (c.ratio - beta), bb = c.mu2 * beta;
        double* Ct = C_tang + i * (D * D);
        for (int r = 0; r < D; ++r)
            for (int q = 0; q < D; q += 2) {
                const double v0 = ((r < 3 && q < 3) ? c.lmbda : 0.0) + (r == q ? c.mu2 : 0.0) - a * (nrm[r] * nrm[q]) -
                                  bb * ((r == q ? 1.0 : 0.0) - ((r < 3 && q < 3) ? 1.0 / 3.0 : 0.0));
                const double v1 = ((r < 3 && q + 1 < 3) ? c.lmbda : 0.0) + (r == q + 1 ? c.mu2 : 0.0) - a * (nrm[r] * nrm[q + 1]) -
                                  bb * ((r == q + 1 ? 1.0 : 0.0) - ((r < 3 && q + 1 < 3) ? 1.0 / 3.0 : 0.0));
                if (stream) _mm_stream_pd(Ct + r * D + q, _mm_set_pd(v1, v0));
                else { Ct[r * D + q] = v0; Ct[r * D + q + 1] = v1; }
            }
    }
    if (stream) _mm_sfence();
}

