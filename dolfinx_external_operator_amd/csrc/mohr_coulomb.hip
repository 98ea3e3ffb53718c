// mohr_coulomb.hip — Mohr-Coulomb (Abbo-Sloan smoothed) return mapping + forward-mode-through-Newton
// tangent, one lane per quadrature point. The per-point algorithm is csrc/mc_core.h (closed-form
// restatement of the reference's nested jacfwd, doc/demo/demo_plasticity_mohr_coulomb.py:282-555).
//
// Roofline: NOT HBM. Algorithmic traffic is 224 B/point (read 4+4, write 16+4 doubles; +28 B with the
// four diagnostics) against ~3-8 kflop of fp64 per Newton iteration and 1 (elastic) / 3-10 (plastic)
// iterations: FP64-VALU- and divergence-bound. HBM GB/s is still reported by the bench.
//
// Memory side: deps / sigma_n are 32 B per point (two 16-byte loads per lane, both halves of every
// 128-byte line are consumed by the same wave). C_tang (128 B per point) and sigma go through a
// wave-private LDS slice so every global store instruction writes 1 KiB of consecutive bytes; the
// diagnostics are naturally lane-linear.
//
// Divergence: like vmap(while_loop) (:564-571) a wave iterates until its slowest lane has converged;
// converged lanes are masked off, so per-point results equal the scalar algorithm.
#include "dxo_common.h"
#include "mc_core.h"

namespace {

constexpr int MC_ROW = 18;  // LDS doubles per point: 16 C_tang + 2 pad (144 B stride: conflict-free b128 writes)

template <bool NT>
__device__ __forceinline__ void st16(dxo_f64x2* p, dxo_f64x2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__global__ __launch_bounds__(DXO_BLOCK) void mc_point(mc::Const k, int64_t n, const double* __restrict__ deps,
                                                      const double* __restrict__ sigma_n, double* __restrict__ C_tang,
                                                      double* __restrict__ sigma, int32_t* __restrict__ niter,
                                                      double* __restrict__ yielding, double* __restrict__ norm_res,
                                                      double* __restrict__ dlambda) {
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * DXO_WAVE * MC_ROW];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* X = lds + wave * (DXO_WAVE * MC_ROW);
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t p0 = tile * DXO_WAVE;
        const int npts = (n - p0 < DXO_WAVE) ? (int)(n - p0) : DXO_WAVE;
        const bool live = lane < npts;
        const int64_t i = p0 + (live ? lane : 0);
        const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + i * 4);
        const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + i * 4);
        const dxo_f64x2 e01 = ge[0], e23 = ge[1], s01 = gs[0], s23 = gs[1];
        const double e[4] = {e01.x, e01.y, e23.x, e23.y};
        const double s[4] = {s01.x, s01.y, s23.x, s23.y};
        mc::Result R;
        mc::return_map(k, e, s, R);
        // ---- outputs through the wave's LDS slice: point-per-lane rows -> lane-linear 16-byte stores
#pragma unroll
        for (int c = 0; c < 8; ++c) X2[(lane * MC_ROW) / 2 + c] = dxo_f64x2{R.C_tang[2 * c], R.C_tang[2 * c + 1]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + p0 * 16);
        const int nct = npts * 8;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = it * DXO_WAVE + lane;
            const int pt = q >> 3, c = q & 7;
            const dxo_f64x2 v = X2[(pt * MC_ROW) / 2 + c];
            if (q < nct) st16<true>(gc + q, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        X2[lane * 2] = dxo_f64x2{R.sigma[0], R.sigma[1]};
        X2[lane * 2 + 1] = dxo_f64x2{R.sigma[2], R.sigma[3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + p0 * 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * DXO_WAVE + lane;
            if (q < npts * 2) st16<true>(gg + q, X2[q]);
        }
        if (live) {
            if (niter) niter[i] = R.niter;
            if (yielding) yielding[i] = R.yielding;
            if (norm_res) norm_res[i] = R.norm_res;
            if (dlambda) dlambda[i] = R.dlambda;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

struct McLaunch {
    mc::Const k;
    bool d_niter, d_yield, d_res, d_dl;
};

int mc_launch(dxo_ctx* ctx, const McLaunch& L, int64_t n, const double* deps, const double* sigma_n, double* C_tang,
              double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda, hipStream_t s) {
    if (n == 0) return DXO_OK;
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
    hipLaunchKernelGGL(mc_point, dim3(grid), dim3(DXO_BLOCK), 0, s, L.k, n, deps, sigma_n, C_tang, sigma, niter, yielding,
                       norm_res, dlambda);
    return DXO_OK;
}

int mc_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const McLaunch& L = *static_cast<const McLaunch*>(user);
    int o = 2;
    int32_t* it = L.d_niter ? (int32_t*)d_out[o++] : nullptr;
    double* yl = L.d_yield ? (double*)d_out[o++] : nullptr;
    double* nr = L.d_res ? (double*)d_out[o++] : nullptr;
    double* dl = L.d_dl ? (double*)d_out[o++] : nullptr;
    return mc_launch(ctx, L, m, (const double*)d_in[0], (const double*)d_in[1], (double*)d_out[0], (double*)d_out[1], it,
                     yl, nr, dl, s);
}

}  // namespace

extern "C" int dxo_mohr_coulomb(dxo_ctx* ctx, const dxo_mc_params* prm, int64_t n, int mem, const double* deps,
                                const double* sigma_n, double* C_tang, double* sigma, int32_t* niter, double* yielding,
                                double* norm_res, double* dlambda) {
    if (!ctx) return DXO_E_NULL;
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb: params is NULL");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mohr_coulomb: bad mem");
    if (n > 0 && (!deps || !sigma_n || !C_tang || !sigma)) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb: NULL array");
    if (prm->nitermax < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb: nitermax < 0");
    const uintptr_t a16 = (uintptr_t)deps | (uintptr_t)sigma_n | (uintptr_t)C_tang | (uintptr_t)sigma;
    const uintptr_t a8 = (uintptr_t)yielding | (uintptr_t)norm_res | (uintptr_t)dlambda;
    if (mem == DXO_MEM_DEVICE && (a16 & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb: deps/sigma_n/C_tang/sigma must be 16-byte aligned");
    if ((a16 & 7u) || (a8 & 7u) || ((uintptr_t)niter & 3u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb: misaligned array");
    McLaunch L{mc::make_const(prm->E, prm->nu, prm->c, prm->phi, prm->psi, prm->theta_T, prm->a, prm->tol, prm->nitermax),
               niter != nullptr, yielding != nullptr, norm_res != nullptr, dlambda != nullptr};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = mc_launch(ctx, L, n, deps, sigma_n, C_tang, sigma, niter, yielding, norm_res, dlambda, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{deps, nullptr, 4 * sd}, {sigma_n, nullptr, 4 * sd}};
    std::vector<dxo_span> out = {{nullptr, C_tang, 16 * sd}, {nullptr, sigma, 4 * sd}};
    if (niter) out.push_back({nullptr, niter, sizeof(int32_t)});
    if (yielding) out.push_back({nullptr, yielding, sd});
    if (norm_res) out.push_back({nullptr, norm_res, sd});
    if (dlambda) out.push_back({nullptr, dlambda, sd});
    return dxo_run_host_pipeline(ctx, n, in, out, mc_chunk, &L);
}
