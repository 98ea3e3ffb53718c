// vm_core.h — device code of the von Mises kernels shared between translation units (von_mises.hip, vm_field.hip).
// Reference: doc/demo/demo_plasticity_von_mises.py:185-204 (constants), :307-326 (per-point `_kernel`).
#pragma once

#include "dxo_common.h"

namespace {

struct VmConst {
    double lmbda, mu2, mu3;   // lambda, 2 mu, 3 mu
    double sigma_0, H;
    double mu3_H;             // 3 mu + H
    double ratio;             // 3 mu / (3 mu + H)
    int mark_indeterminate;   // 1: a point with f_elastic == 0 EXACTLY (the reference's 0/0, :318) returns dp = -0.0, so a
                              // consumer that rebuilds the tangent from (sigma, dp) can reproduce the NaN tangent
    int _pad;
};

VmConst make_const(const dxo_vm_params& p) {
    VmConst c;
    // demo_plasticity_von_mises.py:190-191
    c.lmbda = p.E * p.nu / (1.0 + p.nu) / (1.0 - 2.0 * p.nu);
    const double mu = p.E / 2.0 / (1.0 + p.nu);
    c.mu2 = 2.0 * mu;
    c.mu3 = 3 * mu;
    c.sigma_0 = p.sigma_0;
    c.H = p.H;
    c.mu3_H = 3 * mu + p.H;
    c.ratio = 3 * mu / (3 * mu + p.H);
    c.mark_indeterminate = 0;
    c._pad = 0;
    return c;
}

// Per-point radial return. Outputs: sigma[D], dp, direction n[D], and the two scalars of the
// tangent C_tang = C_elas - a n(x)n - b dev  (a = 3mu(3mu/(3mu+H) - beta), b = 2 mu beta).
template <int D>
__device__ __forceinline__ void vm_return_map(const VmConst& c, const double (&deps)[D], const double (&sn)[D],
                                              double p, double (&sig)[D], double& dp, double (&nrm)[D],
                                              double& a, double& b) {
    // sigma_elastic = sigma_n + C_elas @ deps  (:309); C_elas = lmbda * 1(x)1 + 2 mu I on the Mandel vector
    const double tr_e = deps[0] + deps[1] + deps[2];
    double se[D];
#pragma unroll
    for (int i = 0; i < D; ++i) se[i] = sn[i] + ((i < 3 ? c.lmbda * tr_e : 0.0) + c.mu2 * deps[i]);
    // s = deviatoric @ sigma_elastic (:310)
    const double mean = (se[0] + se[1] + se[2]) * (1.0 / 3.0);
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = i < 3 ? se[i] - mean : se[i];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) ss += s[i] * s[i];
    const double sigma_eq = sqrt(3.0 / 2.0 * ss);                 // :311
    const double f_el = sigma_eq - c.sigma_0 - c.H * p;          // :313
    // :314 f_plus = (f + sqrt(f^2)) / 2. sqrt(fl(f*f)) == |f| exactly for every double whose square neither overflows
    // nor underflows, so |f| is used (f is of the order of sigma_0)
    const double f_plus = (f_el + fabs(f_el)) / 2.0;
    dp = f_plus / c.mu3_H;                                       // :316
    // n_elas = s / sigma_eq * f_plus / f_el (:318) and beta = 3 mu dp / sigma_eq (:319): the two quotients 1/sigma_eq and
    // f_plus/f_el are formed ONCE and multiplied in (an fp64 division is ~11 VALU instructions; the reference's form has
    // 2 per component + 1 = 13 of them, which was a third of this function). Same special values as the reference:
    // f_el == 0 gives 0 * inf = NaN in every n (0/0 there), sigma_eq == 0 gives 0 * inf = NaN in n and beta, an elastic
    // point gives n = +-0 and beta = 0. Finite results differ from the two-division form by at most a rounding each.
    const double r_eq = 1.0 / sigma_eq;
    const double ratio_f = f_plus / f_el;
    const double beta = c.mu3 * dp * r_eq;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        nrm[i] = s[i] * r_eq * ratio_f;
        sig[i] = se[i] - beta * s[i];                            // :321
    }
    a = c.mu3 * (c.ratio - beta);                                // :324
    b = c.mu2 * beta;
    if (c.mark_indeterminate && f_el == 0.0) dp = -0.0;          // value unchanged (0), sign bit = "n_elas was 0/0"
}

// The tangent's state (n, a, b) from the RETURNED (sigma, dp) alone (derivation: the comment above vm_expand_point in von_mises.hip).
// Shared by the tangent rebuild (dxo_vm_expand_tangent) and by the state-based matrix-free tangent action (adjoint.hip).
template <int D>
__device__ __forceinline__ void vm_tangent_state(const VmConst& c, const double (&sig)[D], double dp,
                                                 double (&nrm)[D], double& a, double& b) {
    const double mean = (sig[0] + sig[1] + sig[2]) * (1.0 / 3.0);
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = i < 3 ? sig[i] - mean : sig[i];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) ss += s[i] * s[i];
    const double sigma_eq = sqrt(3.0 / 2.0 * ss);
    const double beta = c.mu3 * dp / (sigma_eq + c.mu3 * dp);
    const double ind = dp > 0.0 ? 1.0 : 0.0;
    const bool marked = dp == 0.0 && __builtin_signbit(dp);   // the producer's mark for f_elastic == 0 (vm_core.h)
#pragma unroll
    for (int i = 0; i < D; ++i) nrm[i] = s[i] / sigma_eq * ind;
    a = marked ? __builtin_nan("") : c.mu3 * (c.ratio - beta);   // NaN * (n_i n_j) = NaN in every entry, as in the reference
    b = c.mu2 * beta;
}

// t = C_tang e for C_tang = C_elas - a n(x)n - b dev WITHOUT forming the matrix: 56 bytes of state per point instead of the 288 of
// the d = 6 tangent block, ~40 flops instead of 36 loads + 36 FMAs.
template <int D>
__device__ __forceinline__ void vm_tangent_times(const VmConst& c, const double (&nrm)[D], double a, double b, const double (&e)[D],
                                                 double (&t)[D]) {
    const double tr = e[0] + e[1] + e[2];
    double ne = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) ne += nrm[i] * e[i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const double vol = i < 3 ? tr : 0.0;
        t[i] = (c.lmbda * vol + c.mu2 * e[i]) - a * (nrm[i] * ne) - b * (e[i] - vol * (1.0 / 3.0));
    }
}

// Entry (i, j) of C_elas and of `deviatoric` on the Mandel vector (:193-204).
__device__ __forceinline__ double c_elas_ij(const VmConst& c, int i, int j) {
    return ((i < 3 && j < 3) ? c.lmbda : 0.0) + (i == j ? c.mu2 : 0.0);
}
__device__ __forceinline__ double dev_ij(int i, int j) {
    return (i == j ? 1.0 : 0.0) - ((i < 3 && j < 3) ? 1.0 / 3.0 : 0.0);
}

// ------------------------------------------------------------------ wave-tile helpers
__device__ __forceinline__ void wave_lds_fence() {
    // Orders this wave's LDS traffic for the compiler; the hardware already executes one wave's DS
    // instructions in issue order, so no s_barrier and no cross-wave wait is involved.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool NT>
__device__ __forceinline__ void store16(dxo_f64x2* ptr, dxo_f64x2 v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, ptr);
    else
        *ptr = v;
}
template <bool NT>
__device__ __forceinline__ void store8(double* ptr, double v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, ptr);
    else
        *ptr = v;
}

template <int D>
struct VmTile {
    static constexpr int PTS = DXO_WAVE;            // points per wave tile
    static constexpr int CH_VEC = D / 2;            // 16-byte chunks per lane for one [PTS][D] block
    static constexpr int CH_CT = D * D / 2;         // 16-byte chunks per point of C_tang = per lane per tile
    static constexpr int ST = D + 2;                // LDS state doubles per point: n[D], a, b
    static constexpr int X_DOUBLES = PTS * D;       // deps staging, later sigma staging
    static constexpr int Y_DOUBLES = PTS * (D > ST ? D : ST);  // sigma_n staging, later state
    static constexpr int WAVE_DOUBLES = X_DOUBLES + Y_DOUBLES;
    static constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
};

// Phase C of a wave tile: the 64 points' tangent state (n[D], a, b per point) sits in the wave's LDS slice Y;
// lanes walk the tile's C_tang block in OUTPUT order (16-byte chunk q = it*64 + lane), so every store
// instruction of the wave covers 1 KiB of consecutive addresses.
// The chunk's coordinates — point pt = q / CH_CT, row i and column pair j0 inside the point's D x D block — are
// carried from one iteration to the next (q grows by 64 = A * CH_CT + R: k += R with a carry into pt) instead of
// being divided out of q each time, and C_elas / deviatoric enter through four 0/1 flags
//   e = [i < 3 and j < 3],  d = [i == j]:   C_elas(i,j) = e lambda + d 2mu,   dev(i,j) = d - e/3
// (the same values as c_elas_ij / dev_ij, so elastic points still return C_elas bit for bit). Round 1's form
// (divisions by 18 and 3, nested selects) cost 54 VALU instructions per 16-byte store — 970 of the ~1 300 per tile,
// which is what made the fused strain + return-map kernel VALU-bound (profiles: 2 250 VALU instructions per tile).
// Tried and dropped for d = 6: three whole points per store instruction (54 active lanes, a lane's chunk fixed for the
// tile: ~10 VALU per store) — the 864-byte stores leave partial 128-byte lines and vm_tile lost 8 % (4 870 vs 5 270
// GB/s on the same box), so every store instruction keeps covering 1 KiB.
// FULL: the tile has all 64 points — no per-lane guard, hence no exec-mask branch around each store and the scheduler
// may run the LDS reads of one iteration under the arithmetic of the previous one.
template <int D, bool NT, bool FULL = false>
__device__ __forceinline__ void vm_store_tangent(const VmConst& c, const double* Y, dxo_f64x2* g_c, int nct, int lane) {
    using T = VmTile<D>;
    const dxo_f64x2* Y2 = reinterpret_cast<const dxo_f64x2*>(Y);
    constexpr int A = DXO_WAVE / T::CH_CT, R = DXO_WAVE % T::CH_CT;
    int pt = lane / T::CH_CT;               // chunk q = lane of iteration 0
    int k = lane - pt * T::CH_CT;           // chunk inside the point's block
    int q = lane;
    // partial unroll: a full unroll lets the scheduler hoist all 3*CH_CT LDS reads and spill
#pragma unroll T::CH_VEC
    for (int it = 0; it < T::CH_CT; ++it) {
        const int i = D == 6 ? (k * 11) >> 5 : k / T::CH_VEC;   // k / 3 for k < 18
        const int j0 = (k - i * T::CH_VEC) * 2;                 // first of two columns
        const double n_i = Y[pt * T::ST + i];
        const dxo_f64x2 n_j = Y2[pt * (T::ST / 2) + (j0 >> 1)];
        const dxo_f64x2 ab = Y2[pt * (T::ST / 2) + T::CH_VEC];
        const double e0 = (i < 3 && j0 < 3) ? 1.0 : 0.0, e1 = (i < 3 && j0 + 1 < 3) ? 1.0 : 0.0;
        const double d0 = i == j0 ? 1.0 : 0.0, d1 = i == j0 + 1 ? 1.0 : 0.0;
        dxo_f64x2 out;
        out.x = (e0 * c.lmbda + d0 * c.mu2) - ab.x * (n_i * n_j.x) - ab.y * (d0 - e0 * (1.0 / 3.0));
        out.y = (e1 * c.lmbda + d1 * c.mu2) - ab.x * (n_i * n_j.y) - ab.y * (d1 - e1 * (1.0 / 3.0));
        if (FULL || q < nct) store16<NT>(g_c + q, out);
        q += DXO_WAVE;
        k += R;
        const bool carry = k >= T::CH_CT;
        pt += carry ? A + 1 : A;
        k -= carry ? T::CH_CT : 0;
    }
}


// Phase C, row form: every lane builds the D x D tangent of ITS OWN point in registers (the 0/1 patterns of C_elas and of the
// deviatoric projector are compile-time constants per entry: two FMAs and a shared product n_i n_j each) and the wave turns the
// point-per-lane rows into output order through LDS, half a tile (32 points) at a time so the staging block stays at
// 32 * D*D doubles. Same arithmetic per entry as vm_store_tangent, but ~90 instead of ~650 vector instructions per lane and
// tile at d = 6 (that walk pays index carries, flag selects and three LDS reads for every 16-byte store); the price is
// D*D/2 ds_write_b128 + D*D/2 ds_read_b128 per lane. W: the wave's LDS slice, at least VM_ROWS_DOUBLES<D> doubles, free on entry.
template <int D>
constexpr int VM_ROWS_STRIDE = (D * D / 2) % 4 == 0 ? D * D / 2 + 1 : D * D / 2;   // 16-byte chunks per staged point (odd multiple of 4 banks)
template <int D>
constexpr int VM_ROWS_DOUBLES = 32 * VM_ROWS_STRIDE<D> * 2;

template <int D, bool NT>
__device__ __forceinline__ void vm_store_tangent_rows(const VmConst& c, double* W, const double (&nrm)[D], double a, double b,
                                                      dxo_f64x2* g_c, int nct, int lane) {
    constexpr int CT = D * D / 2, HV = D / 2, RS = VM_ROWS_STRIDE<D>;
    dxo_f64x2* S2 = reinterpret_cast<dxo_f64x2*>(W);
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        // the half's 32 lanes build and stage their rows one at a time (D/2 chunks live, not D*D/2: the kernels that call this
        // sit at their register budget); the other half idles through the branch
        if ((lane >> 5) == h) {
            double ah = a, bh = b;
            double nh[D];
#pragma unroll
            for (int i = 0; i < D; ++i) {
                nh[i] = nrm[i];
                asm volatile("" : "+v"(nh[i]));
            }
            asm volatile("" : "+v"(ah), "+v"(bh));     // not loop-invariant: keeps the D*D entries from being hoisted (and held) across the halves
#pragma unroll
            for (int i = 0; i < D; ++i)
#pragma unroll
                for (int jj = 0; jj < HV; ++jj) {
                    const int j0 = 2 * jj;
                    const double e0 = (i < 3 && j0 < 3) ? 1.0 : 0.0, e1 = (i < 3 && j0 + 1 < 3) ? 1.0 : 0.0;
                    const double d0 = i == j0 ? 1.0 : 0.0, d1 = i == j0 + 1 ? 1.0 : 0.0;
                    dxo_f64x2 out;
                    out.x = (e0 * c.lmbda + d0 * c.mu2) - ah * (nh[i] * nh[j0]) - bh * (d0 - e0 * (1.0 / 3.0));
                    out.y = (e1 * c.lmbda + d1 * c.mu2) - ah * (nh[i] * nh[j0 + 1]) - bh * (d1 - e1 * (1.0 / 3.0));
                    S2[(lane & 31) * RS + i * HV + jj] = out;
                }
        }
        wave_lds_fence();
#pragma unroll 3
        for (int it = 0; it < CT / 2; ++it) {
            const int q = it * DXO_WAVE + lane;                 // chunk inside the half tile, output order
            const int src = RS == CT ? q : (q / CT) * RS + (q % CT);
            const int gq = h * 32 * CT + q;
            if (gq < nct) store16<NT>(g_c + gq, S2[src]);
        }
        wave_lds_fence();
    }
}


}  // namespace
