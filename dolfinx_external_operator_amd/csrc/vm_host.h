// vm_host.h — the host half of DXO_MEM_HOST dxo_von_mises with option "vm_host_tangent" = 1: consistent tangent from
// the RETURNED state (sigma, dp) on the CPU (plain C++17 + SSE2, no HIP types: also built on its own for the CPU tests
// and the sanitizers). Same formulas and operation order as vm_tangent_state / vm_store_tangent (von_mises.hip),
// reference: demo_plasticity_von_mises.py:318-324 with s = dev sigma.
#pragma once

#include <emmintrin.h>
#include <immintrin.h>

#include <cmath>
#include <cstdint>
#include <limits>

struct VmHostConst {
    double lmbda, mu2, mu3;   // lambda, 2 mu, 3 mu
    double ratio;             // 3 mu / (3 mu + H)
};

// portable form (SSE2 stores): every x86-64 CPU
template <int D>
void vm_host_rebuild_range_sse2(const VmHostConst& c, const double* __restrict__ sigma, double* __restrict__ dp,
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


// AVX2 + FMA form, chosen at run time (vm_host_rebuild_range below): the host half shares its CPUs with the caller (a
// container's CPU quota caps the worker threads, host_pool.h), so per-thread speed is what the rebuild rate follows.
// A point's D x D block is written as 32-byte vectors in FLAT order. For D = 6 the (row, column) pattern of the nine
// vectors repeats every two rows:
//   [r,0..3]   [r,4 r,5 r+1,0 r+1,1]   [r+1,2..5]
// so each vector is  C_elas[k] - b dev[k] - a (nr[k] * nq[k])  with nr, nq built from three loads of n. Same formulas as
// the portable form; products may be fused, so results agree with it to rounding (tests/test_host_half.py).
template <int D>
__attribute__((target("avx2,fma"))) void vm_host_rebuild_range_avx2(const VmHostConst& c, const double* __restrict__ sigma,
                                                                    double* __restrict__ dp, double* __restrict__ C_tang,
                                                                    int64_t b, int64_t e) {
    constexpr int NV = D * D / 4;
    alignas(32) double ce[D * D], dv[D * D];
    for (int r = 0; r < D; ++r)
        for (int q = 0; q < D; ++q) {
            const bool vol = r < 3 && q < 3;
            ce[r * D + q] = (vol ? c.lmbda : 0.0) + (r == q ? c.mu2 : 0.0);
            dv[r * D + q] = (r == q ? 1.0 : 0.0) - (vol ? 1.0 / 3.0 : 0.0);
        }
    __m256d CE[NV], DV[NV];
    for (int k = 0; k < NV; ++k) {
        CE[k] = _mm256_load_pd(ce + 4 * k);
        DV[k] = _mm256_load_pd(dv + 4 * k);
    }
    const bool stream = (((uintptr_t)C_tang) & 31u) == 0;   // D*D*8 is a multiple of 32: every point block is aligned
    for (int64_t i = b; i < e; ++i) {
        const double* sg = sigma + i * D;
        double* Ct = C_tang + i * (D * D);
        const double dpi = dp[i];
        if (dpi == 0.0 && std::signbit(dpi)) {   // the kernel's mark for f_elastic == 0 (see the portable form)
            dp[i] = 0.0;
            for (int k = 0; k < D * D; ++k) Ct[k] = std::numeric_limits<double>::quiet_NaN();
            continue;
        }
        const double mean = (sg[0] + sg[1] + sg[2]) * (1.0 / 3.0);
        alignas(32) double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < D; ++k) s[k] = k < 3 ? sg[k] - mean : sg[k];
        double ss = 0.0;
        for (int k = 0; k < D; ++k) ss += s[k] * s[k];
        const double sigma_eq = std::sqrt(3.0 / 2.0 * ss);
        const double beta = c.mu3 * dpi / (sigma_eq + c.mu3 * dpi);
        const double ind = dpi > 0.0 ? 1.0 : 0.0;
        const double a = c.mu3 * (c.ratio - beta), bb = c.mu2 * beta;
        // n = s / sigma_eq * ind, two vector divisions instead of D scalar ones
        const __m256d veq = _mm256_set1_pd(sigma_eq), vind = _mm256_set1_pd(ind);
        alignas(32) double n[8];
        _mm256_store_pd(n, _mm256_mul_pd(_mm256_div_pd(_mm256_load_pd(s), veq), vind));
        _mm256_store_pd(n + 4, _mm256_mul_pd(_mm256_div_pd(_mm256_load_pd(s + 4), veq), vind));
        const __m256d va = _mm256_set1_pd(a), vb = _mm256_set1_pd(bb);
        __m256d out[NV];
        if constexpr (D == 6) {
            const __m256d n0123 = _mm256_load_pd(n), n2345 = _mm256_loadu_pd(n + 2);
            const __m256d n4501 = _mm256_set_pd(n[1], n[0], n[5], n[4]);
            for (int rp = 0; rp < 3; ++rp) {     // rows 2 rp, 2 rp + 1
                const double r0 = n[2 * rp], r1 = n[2 * rp + 1];
                const __m256d a0 = _mm256_mul_pd(va, _mm256_set1_pd(r0)), a1 = _mm256_mul_pd(va, _mm256_set1_pd(r1));
                const __m256d a01 = _mm256_mul_pd(va, _mm256_set_pd(r1, r1, r0, r0));
                out[3 * rp + 0] = _mm256_fnmadd_pd(a0, n0123, _mm256_fnmadd_pd(vb, DV[3 * rp + 0], CE[3 * rp + 0]));
                out[3 * rp + 1] = _mm256_fnmadd_pd(a01, n4501, _mm256_fnmadd_pd(vb, DV[3 * rp + 1], CE[3 * rp + 1]));
                out[3 * rp + 2] = _mm256_fnmadd_pd(a1, n2345, _mm256_fnmadd_pd(vb, DV[3 * rp + 2], CE[3 * rp + 2]));
            }
        } else {
            const __m256d n0123 = _mm256_load_pd(n);
            for (int r = 0; r < 4; ++r)
                out[r] = _mm256_fnmadd_pd(_mm256_mul_pd(va, _mm256_set1_pd(n[r])), n0123, _mm256_fnmadd_pd(vb, DV[r], CE[r]));
        }
        if (stream)
            for (int k = 0; k < NV; ++k) _mm256_stream_pd(Ct + 4 * k, out[k]);
        else
            for (int k = 0; k < NV; ++k) _mm256_storeu_pd(Ct + 4 * k, out[k]);
    }
    if (stream) _mm_sfence();
}

template <int D>
void vm_host_rebuild_range(const VmHostConst& c, const double* __restrict__ sigma, double* __restrict__ dp,
                           double* __restrict__ C_tang, int64_t b, int64_t e) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (avx2) vm_host_rebuild_range_avx2<D>(c, sigma, dp, C_tang, b, e);
    else vm_host_rebuild_range_sse2<D>(c, sigma, dp, C_tang, b, e);
}
