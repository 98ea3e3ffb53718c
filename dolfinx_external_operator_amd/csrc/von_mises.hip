// von_mises.hip — von Mises radial return + consistent tangent, fused (C_tang, sigma, dp).
//
// Algorithm: the reference's per-point `_kernel`, doc/demo/demo_plasticity_von_mises.py:307-326
// (constants :185-204, batched caller return_mapping :298-332, wrapper C_tang_impl :343-352).
//
// Roofline: HBM. Algorithmic traffic per quadrature point (fp64):
//   d = 4: read 4+4+1, write 16+4+1 doubles = 240 B      d = 6: read 6+6+1, write 36+6+1 = 448 B
// against ~150 flop — < 1 flop/B, so the design goal is: every global access a fully coalesced
// 16 B/lane (1 KiB per wave-instruction) transaction, nothing read or written twice.
//
// Two kernels:
//   vm_point  (variant 0)  one lane = one point, strided 8-byte AoS accesses. Needs only 8-byte
//                          alignment; the fallback and the on-GPU cross-check of variant 1.
//   vm_tile   (variant 1)  one WAVE = one tile of 64 consecutive points.
//     A  the tile's deps / sigma_n blocks (64*d contiguous doubles each) are fetched with
//        lane-linear 16-byte loads, parked in the wave's private LDS slice and re-read
//        point-per-lane (an AoS->per-lane transpose through LDS).
//     B  each lane does the radial return of its own point in registers.
//     C  sigma goes back through LDS to lane-linear 16-byte stores. For the d*d tangent the
//        wave does NOT transpose 64*d*d doubles through LDS: each lane leaves only
//        (n[0..d), a, b) — d+2 doubles — in LDS, and the wave then walks the tile's C_tang block
//        in OUTPUT order: chunk q = it*64 + lane covers entries (pt, i, j0), (pt, i, j0+1) with
//        pt = q / (d*d/2); the lane rebuilds both from C_elas(i,j) - a n_i n_j - b dev(i,j).
//        LDS per wave: 7 KiB (d=6) / 5 KiB (d=4) instead of 18 KiB, so 16+ waves stay resident
//        per CU, and every C_tang store instruction writes 1 KiB of consecutive bytes.
//   No cross-wave communication, hence no __syncthreads(): LDS slices are wave-private and DS
//   operations of one wave execute in order; only the compiler must be kept from reordering.
#include <chrono>

#include "dxo_common.h"
#include "vm_core.h"
#include "vm_host.h"

#ifndef DXO_VM_NT_LOADS
#define DXO_VM_NT_LOADS 0    // non-temporal input loads (scripts/exp/vmtile_ab.py)
#endif
#ifndef DXO_VM_FULL_PATH
#define DXO_VM_FULL_PATH 1   // guard-free body for tiles of 64 points
#endif
#ifndef DXO_VM_MIN_BLOCKS
#define DXO_VM_MIN_BLOCKS 2  // __launch_bounds__ second argument of vm_tile: 70-90 VGPRs either way; the schedule under 2 was 4 % faster
                             // on a slow-clocked box (5 450 vs 5 245 GB/s) and equal (+-0.3 %) on the others (scripts/exp/vmtile_ab.py)
#endif

namespace {

// ------------------------------------------------------------------ variant 0: lane = point
template <int D>
__global__ __launch_bounds__(DXO_BLOCK) void vm_point(VmConst c, int64_t n, const double* __restrict__ deps,
                                                      const double* __restrict__ sigma_n,
                                                      const double* __restrict__ p, double* __restrict__ C_tang,
                                                      double* __restrict__ sigma, double* __restrict__ dp_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double e[D], sn[D], sig[D], nrm[D], dp, a, b;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            e[k] = deps[i * D + k];
            sn[k] = sigma_n[i * D + k];
        }
        vm_return_map<D>(c, e, sn, p[i], sig, dp, nrm, a, b);
#pragma unroll
        for (int k = 0; k < D; ++k) sigma[i * D + k] = sig[k];
        dp_out[i] = dp;
        if (!C_tang) continue;   // (sigma, dp) only: the compact multi-GPU gather rebuilds every tangent from them
        double* Ct = C_tang + i * (D * D);
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int q = 0; q < D; ++q) Ct[r * D + q] = c_elas_ij(c, r, q) - a * (nrm[r] * nrm[q]) - b * dev_ij(r, q);
    }
}

// ------------------------------------------------------------------ variant 1: wave = 64-point tile (helpers: vm_core.h)
// A tile of `npts` points starting at point p0 in three phases; FULL = (npts == 64): every guard folds away.
template <int D>
struct VmTileIn {
    dxo_f64x2 ve[D / 2], vs[D / 2];
    double p;
};

// ---- A1: lane-linear global loads into registers
template <int D, bool FULL>
__device__ __forceinline__ void vm_tile_load(VmTileIn<D>& r, int64_t p0, int npts, int lane, const double* __restrict__ deps,
                                             const double* __restrict__ sigma_n, const double* __restrict__ p) {
    using T = VmTile<D>;
    const int nvec = npts * T::CH_VEC;                              // valid 16-byte chunks of a [npts][D] block
    const dxo_f64x2* g_e = reinterpret_cast<const dxo_f64x2*>(deps + p0 * D);
    const dxo_f64x2* g_s = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
#pragma unroll
    for (int k = 0; k < T::CH_VEC; ++k) {
        const int idx = k * DXO_WAVE + lane;
        const bool ok = FULL || idx < nvec;
#if DXO_VM_NT_LOADS
        r.ve[k] = ok ? __builtin_nontemporal_load(g_e + idx) : dxo_f64x2{0.0, 0.0};
        r.vs[k] = ok ? __builtin_nontemporal_load(g_s + idx) : dxo_f64x2{0.0, 0.0};
#else
        r.ve[k] = ok ? g_e[idx] : dxo_f64x2{0.0, 0.0};
        r.vs[k] = ok ? g_s[idx] : dxo_f64x2{0.0, 0.0};
#endif
    }
    r.p = (FULL || lane < npts) ? p[p0 + lane] : 0.0;
}

// ---- A2: registers -> LDS (lane-linear); the transposed read-back is vm_tile_unstage
template <int D>
__device__ __forceinline__ void vm_tile_stage(const VmTileIn<D>& r, int lane, double* X, double* Y) {
    using T = VmTile<D>;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
#pragma unroll
    for (int k = 0; k < T::CH_VEC; ++k) {
        X2[k * DXO_WAVE + lane] = r.ve[k];
        Y2[k * DXO_WAVE + lane] = r.vs[k];
    }
    wave_lds_fence();
}

template <int D>
__device__ __forceinline__ void vm_tile_unstage(int lane, double* X, double* Y, double (&e)[D], double (&sn)[D]) {
    using T = VmTile<D>;
    const dxo_f64x2* X2 = reinterpret_cast<const dxo_f64x2*>(X);
    const dxo_f64x2* Y2 = reinterpret_cast<const dxo_f64x2*>(Y);
#pragma unroll
    for (int k = 0; k < T::CH_VEC; ++k) {
        const dxo_f64x2 a2 = X2[lane * T::CH_VEC + k];
        const dxo_f64x2 b2 = Y2[lane * T::CH_VEC + k];
        e[2 * k] = a2.x;
        e[2 * k + 1] = a2.y;
        sn[2 * k] = b2.x;
        sn[2 * k + 1] = b2.y;
    }
    wave_lds_fence();  // staging slices are about to be reused
}

// ---- B + C: radial return of this lane's point, then output-ordered coalesced stores
// WT = false: the tangent is not written (C_tang == NULL: a caller that rebuilds it from (sigma, dp), dxo_vm_expand_tangent)
template <int D, bool NT, bool FULL, bool WT = true>
__device__ __forceinline__ void vm_tile_finish(const VmConst& c, const double (&e)[D], const double (&sn)[D], double p_l, int64_t p0,
                                               int npts, int lane, double* X, double* Y, double* __restrict__ C_tang,
                                               double* __restrict__ sigma, double* __restrict__ dp_out) {
    using T = VmTile<D>;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
    const int nvec = npts * T::CH_VEC;
    double sig[D], nrm[D], dp, a, b;
    vm_return_map<D>(c, e, sn, p_l, sig, dp, nrm, a, b);
    // sigma -> X (point-per-lane), state -> Y
#pragma unroll
    for (int k = 0; k < T::CH_VEC; ++k) {
        X2[lane * T::CH_VEC + k] = dxo_f64x2{sig[2 * k], sig[2 * k + 1]};
        if constexpr (WT) Y2[lane * (T::ST / 2) + k] = dxo_f64x2{nrm[2 * k], nrm[2 * k + 1]};
    }
    if constexpr (WT) Y2[lane * (T::ST / 2) + T::CH_VEC] = dxo_f64x2{a, b};
    wave_lds_fence();

    if (FULL || lane < npts) store8<NT>(dp_out + p0 + lane, dp);
    dxo_f64x2* g_o = reinterpret_cast<dxo_f64x2*>(sigma + p0 * D);
#pragma unroll
    for (int k = 0; k < T::CH_VEC; ++k) {
        const int idx = k * DXO_WAVE + lane;
        if (FULL || idx < nvec) store16<NT>(g_o + idx, X2[idx]);
    }
    if constexpr (WT) vm_store_tangent<D, NT, FULL>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
    wave_lds_fence();  // next tile overwrites X / Y
}

template <int D, bool NT, bool FULL, bool WT = true>
__device__ __forceinline__ void vm_tile_body(const VmConst& c, int64_t p0, int npts, int lane, double* X, double* Y,
                                             const double* __restrict__ deps, const double* __restrict__ sigma_n,
                                             const double* __restrict__ p, double* __restrict__ C_tang,
                                             double* __restrict__ sigma, double* __restrict__ dp_out) {
    VmTileIn<D> r;
    vm_tile_load<D, FULL>(r, p0, npts, lane, deps, sigma_n, p);
    vm_tile_stage<D>(r, lane, X, Y);
    double e[D], sn[D];
    vm_tile_unstage<D>(lane, X, Y, e, sn);
    vm_tile_finish<D, NT, FULL, WT>(c, e, sn, r.p, p0, npts, lane, X, Y, C_tang, sigma, dp_out);
}

template <int D, bool NT, bool WT = true>
__global__ __launch_bounds__(DXO_BLOCK, DXO_VM_MIN_BLOCKS) void vm_tile(VmConst c, int64_t n, const double* __restrict__ deps,
                                                     const double* __restrict__ sigma_n,
                                                     const double* __restrict__ p, double* __restrict__ C_tang,
                                                     double* __restrict__ sigma, double* __restrict__ dp_out) {
    using T = VmTile<D>;
    __shared__ __attribute__((aligned(16))) double lds[T::WAVES * T::WAVE_DOUBLES];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* X = lds + wave * T::WAVE_DOUBLES;
    double* Y = X + T::X_DOUBLES;

    const int64_t n_tiles = (n + T::PTS - 1) / T::PTS;
    const int64_t tile_stride = (int64_t)gridDim.x * T::WAVES;
    int64_t tile = (int64_t)blockIdx.x * T::WAVES + wave;
    // Persistent grids deal the tiles round-robin to the waves (consecutive workgroups = consecutive tiles, so the 8 XCDs
    // write neighbouring tiles at any time). Tried and dropped: every XCD sweeping its own contiguous eighth (0.729-0.775 of
    // 8 TB/s against 0.790-0.793) and runs of 4 consecutive tiles per wave (0.746-0.762), three interleaved bench runs each.
    // Tried and dropped (round 2): in persistent grids, requesting the wave's NEXT tile's inputs right after this tile's
    // have been staged in LDS, so that their latency runs under the arithmetic and the stores — 0.781-0.784 against
    // 0.778-0.787 of 8 TB/s in four interleaved bench runs each: the loads are not what the kernel waits for.
    for (; tile < n_tiles; tile += tile_stride) {
        const int64_t p0 = tile * T::PTS;
        const int npts = (n - p0 < T::PTS) ? (int)(n - p0) : T::PTS;  // wave-uniform
#if DXO_VM_FULL_PATH
        if (npts == T::PTS) vm_tile_body<D, NT, true, WT>(c, p0, npts, lane, X, Y, deps, sigma_n, p, C_tang, sigma, dp_out);
        else
#endif
            vm_tile_body<D, NT, false, WT>(c, p0, npts, lane, X, Y, deps, sigma_n, p, C_tang, sigma, dp_out);
    }
}

struct VmLaunch {
    VmConst c;
    int d;
    int shape = -1;   // >= 0: launch shape forced by a calibration (0 one tile per wave, k persistent workgroups per CU)
    // host half of the DXO_MEM_HOST pipeline with option "vm_host_tangent" = 1 (see vm_host_rebuild)
    const double* h_sigma = nullptr;
    double* h_dp = nullptr;
    double* h_C_tang = nullptr;
};

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

int vm_launch(dxo_ctx* ctx, const VmLaunch& L, int64_t n, const double* deps, const double* sigma_n,
              const double* p, double* C_tang, double* sigma, double* dp, hipStream_t s) {
    if (n == 0) return DXO_OK;
    const bool can_tile = aligned16(deps) && aligned16(sigma_n) && aligned16(C_tang) && aligned16(sigma);
    const bool tiled = ctx->vm_variant != 0 && can_tile;
    if (tiled) {
        const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
        int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
        // launch shape: the option when set; else the shape a calibration found best for the arena block written to
        // (dxo_vm_output_alloc); else one tile per wave
        int shape = L.shape;
        if (shape < 0 && ctx->blocks_per_cu == 0 && !ctx->arena.empty()) shape = dxo_arena_tuned_shape(ctx, C_tang ? (const void*)C_tang : (const void*)sigma);
        if (shape > 0) {
            const int64_t cap = (int64_t)ctx->compute_units * shape;
            const int64_t full = (n_tiles + DXO_BLOCK / DXO_WAVE - 1) / (DXO_BLOCK / DXO_WAVE);
            grid = (int)(full < cap ? full : cap);
        } else if (L.shape == 0) {
            const int64_t full = (n_tiles + DXO_BLOCK / DXO_WAVE - 1) / (DXO_BLOCK / DXO_WAVE);
            grid = (int)(full > 0x7fffffff ? 0x7fffffff : full);
        }
        const bool nt = ctx->nontemporal != 0;
        if (!C_tang) {   // (sigma, dp) only
            if (L.d == 4) {
                if (nt) hipLaunchKernelGGL((vm_tile<4, true, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
                else    hipLaunchKernelGGL((vm_tile<4, false, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
            } else {
                if (nt) hipLaunchKernelGGL((vm_tile<6, true, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
                else    hipLaunchKernelGGL((vm_tile<6, false, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
            }
        } else if (L.d == 4) {
            if (nt) hipLaunchKernelGGL((vm_tile<4, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
            else    hipLaunchKernelGGL((vm_tile<4, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
        } else {
            if (nt) hipLaunchKernelGGL((vm_tile<6, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
            else    hipLaunchKernelGGL((vm_tile<6, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
        }
    } else {
        const int grid = dxo_grid_for_tiles(ctx, (n + DXO_BLOCK - 1) / DXO_BLOCK, 1);
        if (L.d == 4) hipLaunchKernelGGL((vm_point<4>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
        else          hipLaunchKernelGGL((vm_point<6>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, deps, sigma_n, p, C_tang, sigma, dp);
    }
    return DXO_OK;
}

int vm_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const VmLaunch& L = *static_cast<const VmLaunch*>(user);
    // host-rebuild mode (option vm_host_tangent): the tangent is rebuilt from (sigma, dp) on the host, so the device writes none — the
    // (sigma, dp)-only form of the kernel: 160 instead of 448 B/point of stores, and on the small path no tangent crosses PCIe
    double* C_tang = L.h_C_tang ? nullptr : (double*)d_out[0];
    return vm_launch(ctx, L, m, (const double*)d_in[0], (const double*)d_in[1], (const double*)d_in[2], C_tang, (double*)d_out[1], (double*)d_out[2], s);
}

// ------------------------------------------------------------------ tangent from (sigma, dp)
// The consistent tangent is a function of the RETURNED state alone: radial return keeps the deviatoric direction
// (dev sigma = (1 - beta) s_tr), so with s = dev sigma, sigma_eq = sqrt(3/2 s.s):
//   sigma_eq_tr = sigma_eq + 3 mu dp,  beta = 3 mu dp / sigma_eq_tr,  n = s / sigma_eq (plastic) or 0 (elastic),
// and C_tang = C_elas - 3mu(3mu/(3mu+H) - beta) n(x)n - 2 mu beta dev as in demo_plasticity_von_mises.py:318-324.
// Used by the multi-GPU gather (sharding.py): ranks exchange (sigma, dp) = (d+1) doubles per point over xGMI
// instead of (d*d+d+1) and rebuild the tangent of the REMOTE blocks here at HBM speed. NaN propagates as in the
// reference (sigma NaN or 0/0 -> every entry NaN). The reference's 0/0 at f_el == 0 EXACTLY (:318) leaves no trace in the
// VALUES of (sigma, dp): a producer run with option "vm_mark_indeterminate" (every compact gather sets it) returns
// dp = -0.0 there, and the mark comes out as the reference's all-NaN tangent here, exactly as in vm_host.h. Without the
// mark such a point is rebuilt as C_elas.
template <int D>
__global__ __launch_bounds__(DXO_BLOCK) void vm_expand_point(VmConst c, int64_t n, const double* __restrict__ sigma,
                                                             const double* __restrict__ dp_in,
                                                             double* __restrict__ C_tang) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double sig[D], nrm[D], a, b;
#pragma unroll
        for (int k = 0; k < D; ++k) sig[k] = sigma[i * D + k];
        vm_tangent_state<D>(c, sig, dp_in[i], nrm, a, b);
        double* Ct = C_tang + i * (D * D);
#pragma unroll
        for (int r = 0; r < D; ++r)
#pragma unroll
            for (int q = 0; q < D; ++q) Ct[r * D + q] = c_elas_ij(c, r, q) - a * (nrm[r] * nrm[q]) - b * dev_ij(r, q);
    }
}

template <int D, bool NT>
__global__ __launch_bounds__(DXO_BLOCK, 4) void vm_expand_tile(VmConst c, int64_t n, const double* __restrict__ sigma,
                                                            const double* __restrict__ dp_in,
                                                            double* __restrict__ C_tang) {
    using T = VmTile<D>;
    __shared__ __attribute__((aligned(16))) double lds[T::WAVES * T::WAVE_DOUBLES];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* X = lds + wave * T::WAVE_DOUBLES;
    double* Y = X + T::X_DOUBLES;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
    const int64_t n_tiles = (n + T::PTS - 1) / T::PTS;
    const int64_t tile_stride = (int64_t)gridDim.x * T::WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * T::WAVES + wave; tile < n_tiles; tile += tile_stride) {
        const int64_t p0 = tile * T::PTS;
        const int npts = (n - p0 < T::PTS) ? (int)(n - p0) : T::PTS;
        const int nvec = npts * T::CH_VEC;
        const dxo_f64x2* g_s = reinterpret_cast<const dxo_f64x2*>(sigma + p0 * D);
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const int idx = k * DXO_WAVE + lane;
            X2[idx] = idx < nvec ? g_s[idx] : dxo_f64x2{0.0, 0.0};
        }
        const double dp = lane < npts ? dp_in[p0 + lane] : 0.0;
        wave_lds_fence();
        double sig[D], nrm[D], a, b;
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const dxo_f64x2 v = X2[lane * T::CH_VEC + k];
            sig[2 * k] = v.x;
            sig[2 * k + 1] = v.y;
        }
        vm_tangent_state<D>(c, sig, dp, nrm, a, b);
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) Y2[lane * (T::ST / 2) + k] = dxo_f64x2{nrm[2 * k], nrm[2 * k + 1]};
        Y2[lane * (T::ST / 2) + T::CH_VEC] = dxo_f64x2{a, b};
        wave_lds_fence();
        vm_store_tangent<D, NT>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
        wave_lds_fence();  // next tile overwrites X / Y
    }
}

int vm_expand_launch(dxo_ctx* ctx, const VmLaunch& L, int64_t n, const double* sigma, const double* dp, double* C_tang,
                     hipStream_t s) {
    if (n == 0) return DXO_OK;
    const bool tiled = ctx->vm_variant != 0 && aligned16(sigma) && aligned16(C_tang);
    if (tiled) {
        const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
        int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
        // the tangent stores are vm_tile's: a block calibrated for vm_tile (dxo_vm_output_alloc) lends its launch shape
        const int shape = ctx->blocks_per_cu == 0 && !ctx->arena.empty() ? dxo_arena_tuned_shape(ctx, C_tang) : -1;
        if (shape > 0) {
            const int64_t cap = (int64_t)ctx->compute_units * shape, full = (n_tiles + DXO_BLOCK / DXO_WAVE - 1) / (DXO_BLOCK / DXO_WAVE);
            grid = (int)(full < cap ? full : cap);
        }
        const bool nt = ctx->nontemporal != 0;
        if (L.d == 4) {
            if (nt) hipLaunchKernelGGL((vm_expand_tile<4, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
            else    hipLaunchKernelGGL((vm_expand_tile<4, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
        } else {
            if (nt) hipLaunchKernelGGL((vm_expand_tile<6, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
            else    hipLaunchKernelGGL((vm_expand_tile<6, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
        }
    } else {
        const int grid = dxo_grid_for_tiles(ctx, (n + DXO_BLOCK - 1) / DXO_BLOCK, 1);
        if (L.d == 4) hipLaunchKernelGGL((vm_expand_point<4>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
        else          hipLaunchKernelGGL((vm_expand_point<6>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.c, n, sigma, dp, C_tang);
    }
    return DXO_OK;
}

int vm_expand_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const VmLaunch& L = *static_cast<const VmLaunch*>(user);
    return vm_expand_launch(ctx, L, m, (const double*)d_in[0], (const double*)d_in[1], (double*)d_out[0], s);
}

// ------------------------------------------------------------------ host half: tangent from (sigma, dp) on the CPU
// DXO_MEM_HOST with option "vm_host_tangent" = 1: of the 344 B/point (d = 6) the kernel produces, 288 are the
// tangent, and the tangent is a function of the returned state (vm_tangent_state above). PCIe, not HBM, bounds a
// host call, so only (sigma, dp) cross the link (56 B/point) and the caller's C_tang array is filled by the
// context's host threads from the chunk that has just landed while the next chunks are still in flight. Same
// formulas and operation order as vm_tangent_state / vm_store_tangent; streaming (non-temporal) stores because the
// array is write-only here and larger than any cache. Product code of the host pipeline, not a CPU fallback: the
// return map itself always runs on the GPU.
int vm_host_rebuild(dxo_ctx* ctx, void* user, int64_t first, int64_t m) {
    const VmLaunch& L = *static_cast<const VmLaunch*>(user);
    const int d = L.d;
    const double* sg = L.h_sigma + first * d;
    double* dp = L.h_dp + first;
    double* Ct = L.h_C_tang + first * d * d;
    const VmHostConst hc{L.c.lmbda, L.c.mu2, L.c.mu3, L.c.ratio};
    dxo_host_parallel_for(ctx, m, 512, [&](int64_t b, int64_t e) {
        if (d == 4) vm_host_rebuild_range<4>(hc, sg, dp, Ct, b, e);
        else vm_host_rebuild_range<6>(hc, sg, dp, Ct, b, e);
    });
    return DXO_OK;
}

}  // namespace

extern "C" int dxo_vm_expand_tangent(dxo_ctx* ctx, const dxo_vm_params* prm, int d, int64_t n, int mem,
                                     const double* sigma, const double* dp, double* C_tang) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_expand_tangent: params is NULL");
    if (d != 4 && d != 6) return dxo_fail(ctx, DXO_E_DIM, "dxo_vm_expand_tangent: d must be 4 or 6");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_expand_tangent: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_vm_expand_tangent: bad mem");
    if (n > 0 && (!sigma || !dp || !C_tang)) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_expand_tangent: NULL array");
    if (((uintptr_t)sigma | (uintptr_t)dp | (uintptr_t)C_tang) & 7u)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_vm_expand_tangent: arrays must be 8-byte aligned");
    VmLaunch L{make_const(*prm), d};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = vm_expand_launch(ctx, L, n, sigma, dp, C_tang, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{sigma, nullptr, d * sd}, {dp, nullptr, sd}};
    std::vector<dxo_span> out = {{nullptr, C_tang, d * d * sd}};
    return dxo_run_host_pipeline(ctx, n, in, out, vm_expand_chunk, &L, 1, nullptr, true);
}

extern "C" int dxo_von_mises(dxo_ctx* ctx, const dxo_vm_params* prm, int d, int64_t n, int mem,
                             const double* deps, const double* sigma_n, const double* p, double* C_tang,
                             double* sigma, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises: params is NULL");
    if (d != 4 && d != 6) return dxo_fail(ctx, DXO_E_DIM, "dxo_von_mises: d must be 4 or 6");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_von_mises: bad mem");
    // C_tang may be NULL on the device path: (sigma, dp) only, for callers that rebuild the tangent (dxo_vm_expand_tangent)
    if (n > 0 && (!deps || !sigma_n || !p || (!C_tang && mem != DXO_MEM_DEVICE) || !sigma || !dp))
        return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises: NULL array");
    const uintptr_t all = (uintptr_t)deps | (uintptr_t)sigma_n | (uintptr_t)p | (uintptr_t)C_tang |
                          (uintptr_t)sigma | (uintptr_t)dp;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises: arrays must be 8-byte aligned");
    VmLaunch L{make_const(*prm), d};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        L.c.mark_indeterminate = ctx->vm_mark_indeterminate != 0;
        rc = vm_launch(ctx, L, n, deps, sigma_n, p, C_tang, sigma, dp, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{deps, nullptr, d * sd}, {sigma_n, nullptr, d * sd}, {p, nullptr, sd}};
    if (ctx->vm_host_tangent && n >= ctx->vm_rebuild_min_points) {
        // (sigma, dp) over PCIe, C_tang rebuilt by the host threads from each chunk as it lands (vm_host_rebuild)
        L.h_sigma = sigma;
        L.h_dp = dp;
        L.h_C_tang = C_tang;
        L.c.mark_indeterminate = 1;
        std::vector<dxo_span> out = {{nullptr, nullptr, 0}, {nullptr, sigma, d * sd}, {nullptr, dp, sd}};      // no tangent on the device side (vm_chunk)
        // smaller chunks than the copy mode: the host half of a chunk runs on the calling thread between two enqueues,
        // so the un-overlapped tail of the call is the rebuild of the last DXO_HOST_SLOTS chunks
        const int64_t saved_chunk = ctx->host_chunk_points;
        if (ctx->host_chunk_points > ctx->vm_rebuild_chunk_points) ctx->host_chunk_points = ctx->vm_rebuild_chunk_points;
        const int rc = dxo_run_host_pipeline(ctx, n, in, out, vm_chunk, &L, 1, vm_host_rebuild, true);
        ctx->host_chunk_points = saved_chunk;
        return rc;
    }
    std::vector<dxo_span> out = {{nullptr, C_tang, d * d * sd}, {nullptr, sigma, d * sd}, {nullptr, dp, sd}};
    return dxo_run_host_pipeline(ctx, n, in, out, vm_chunk, &L, 1, nullptr, true);
}

// ------------------------------------------------------------------ output block calibrated with the kernel itself
namespace {
// synthetic inputs of the reference's distribution (deps ~ 3e-3, sigma_n ~ 100, p ~ 1e-3; mostly plastic): values only
// matter in that they are ordinary finite numbers — the kernel is branch-free and HBM-bound
__global__ __launch_bounds__(DXO_BLOCK) void vm_probe_fill(int64_t n_in, int64_t n_d, double* __restrict__ in) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += stride) {
        uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        const double u = (double)(h >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;   // (-1, 1)
        in[i] = i < n_d ? 4e-3 * u : (i < 2 * n_d ? 150.0 * u : 1e-3 * (u < 0 ? -u : u));
    }
}
}  // namespace

extern "C" int dxo_vm_output_alloc(dxo_ctx* ctx, int d, int64_t n, double** C_tang, double** sigma, double** dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!C_tang || !sigma || !dp) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_output_alloc: NULL result pointer");
    *C_tang = *sigma = *dp = nullptr;
    if (d != 4 && d != 6) return dxo_fail(ctx, DXO_E_DIM, "dxo_vm_output_alloc: d must be 4 or 6");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_output_alloc: n < 0");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t sd = sizeof(double);
    const size_t off_s = up((size_t)n * d * d * sd), off_p = off_s + up((size_t)n * d * sd);
    size_t bytes = off_p + up((size_t)n * sd);
    if (bytes == 0) bytes = 256;
    dxo_arena_block blk;
    std::memset(&blk.info, 0, sizeof blk.info);
    blk.info.chosen = -1;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = ctx->stream;
    bool done = false;
    if ((int64_t)bytes >= ctx->placement_min_bytes && ctx->placement_candidates > 1 && ctx->placement_mode >= 1) {
        double* in = nullptr;
        const int64_t n_in = n * (2 * d + 1), n_d = n * d;
        if (hipMalloc((void**)&in, (size_t)n_in * sd) == hipSuccess) {
            hipLaunchKernelGGL(vm_probe_fill, dim3(ctx->compute_units * 8), dim3(DXO_BLOCK), 0, s, n_in, n_d, in);
            const dxo_vm_params ref = {70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0)};   // demo_plasticity_von_mises.py:185-188
            VmLaunch L{make_const(ref), d};
            dxo_arena_probe pr;
            pr.shapes = {0, 32};
            pr.kind = 2;
            pr.bytes_per_launch = (double)n * (double)((2 * d + 1 + d * d + d + 1) * sd);
            pr.good_GBps = 0.0;   // no early exit: the search is the point
            pr.launch = [=](void* p, int shape, hipStream_t st) {
                VmLaunch Ls = L;
                Ls.shape = shape;
                char* b = static_cast<char*>(p);
                (void)vm_launch(ctx, Ls, n, in, in + n_d, in + 2 * n_d, (double*)b, (double*)(b + off_s), (double*)(b + off_p), st);
            };
            done = dxo_arena_alloc_calibrated(ctx, bytes, pr, blk, s);
            (void)hipStreamSynchronize(s);
            (void)hipFree(in);
        } else {
            (void)hipGetLastError();
        }
    }
    if (!done) {
        std::memset(&blk.info, 0, sizeof blk.info);
        blk.info.chosen = -1;
        blk.vmm = nullptr;
        DXO_HIP(ctx, hipMalloc(&blk.ptr, bytes));
        blk.bytes = bytes;
    }
    DXO_HIP(ctx, hipStreamSynchronize(s));
    blk.info.calibration_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    dxo_arena_register(ctx, blk);
    char* b = static_cast<char*>(blk.ptr);
    *C_tang = (double*)b;
    *sigma = (double*)(b + off_s);
    *dp = (double*)(b + off_p);
    return DXO_OK;
}

// ------------------------------------------------------------------ device-resident history variables
// The reference's callback reads sigma_n and p from closure-captured arrays at every call
// (demo_plasticity_von_mises.py:347-348) although they only change at the end of a load step (:564-565). Uploading them
// with every Newton iteration is 56 of the 104 B/point a host call sends (d = 6); with a dxo_vm_state they cross the
// link once, the kernel reads them from HBM, leaves (sigma, dp) of the call beside them, and dxo_vm_state_commit applies
// the load-step update on the device while the caller applies the reference's two statements to its host arrays.
extern "C" int dxo_vm_state_create(dxo_ctx* ctx, int d, int64_t n, dxo_vm_state** out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!out) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_create: out is NULL");
    *out = nullptr;
    if (d != 4 && d != 6) return dxo_fail(ctx, DXO_E_DIM, "dxo_vm_state_create: d must be 4 or 6");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_state_create: n < 0");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t bs = up((size_t)n * d * sizeof(double)), bp = up((size_t)n * sizeof(double));
    void* blob = nullptr;
    DXO_HIP(ctx, hipMalloc(&blob, 2 * bs + 2 * bp + 256));
    dxo_vm_state* st = new dxo_vm_state;
    st->d = d;
    st->n = n;
    st->blob = blob;
    char* b = static_cast<char*>(blob);
    st->sigma_n = reinterpret_cast<double*>(b);
    st->sigma = reinterpret_cast<double*>(b + bs);
    st->p = reinterpret_cast<double*>(b + 2 * bs);
    st->dp = reinterpret_cast<double*>(b + 2 * bs + bp);
    *out = st;
    return DXO_OK;
}

extern "C" void dxo_vm_state_destroy(dxo_ctx* ctx, dxo_vm_state* st) {
    if (!st) return;
    if (ctx) {
        DXO_LOCK(ctx);
        (void)hipSetDevice(ctx->device);
        (void)dxo_ctx_synchronize(ctx);
        if (st->blob) (void)hipFree(st->blob);
    } else if (st->blob) {
        (void)hipFree(st->blob);
    }
    delete st;
}

extern "C" int dxo_vm_state_upload(dxo_ctx* ctx, dxo_vm_state* st, int mem, const double* sigma_n, const double* p) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_upload: state is NULL");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_vm_state_upload: bad mem");
    if (st->n > 0 && (!sigma_n || !p)) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_upload: NULL array");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    const hipMemcpyKind k = mem == DXO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (st->n > 0) {
        DXO_HIP(ctx, hipMemcpyAsync(st->sigma_n, sigma_n, (size_t)st->n * st->d * sizeof(double), k, s));
        DXO_HIP(ctx, hipMemcpyAsync(st->p, p, (size_t)st->n * sizeof(double), k, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));   // the next call may read the mirror from any of the pipeline's streams
    }
    st->uploaded = true;
    st->has_result = false;
    return DXO_OK;
}

extern "C" int dxo_vm_state_download(dxo_ctx* ctx, dxo_vm_state* st, int mem, double* sigma_n, double* p) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_download: state is NULL");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_vm_state_download: bad mem");
    if (!st->uploaded) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_state_download: nothing has been uploaded");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    const hipMemcpyKind k = mem == DXO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (st->n > 0) {
        if (sigma_n) DXO_HIP(ctx, hipMemcpyAsync(sigma_n, st->sigma_n, (size_t)st->n * st->d * sizeof(double), k, s));
        if (p) DXO_HIP(ctx, hipMemcpyAsync(p, st->p, (size_t)st->n * sizeof(double), k, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));
    }
    return DXO_OK;
}

extern "C" int dxo_vm_state_pointers(dxo_ctx* ctx, dxo_vm_state* st, double** sigma_n, double** p, double** sigma, double** dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_pointers: state is NULL");
    if (sigma_n) *sigma_n = st->sigma_n;
    if (p) *p = st->p;
    if (sigma) *sigma = st->sigma;
    if (dp) *dp = st->dp;
    return DXO_OK;
}

extern "C" int dxo_von_mises_state(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_vm_state* st, int mem, const double* deps,
                                   double* C_tang, double* sigma, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !st) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_state: NULL params or state");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_von_mises_state: bad mem");
    if (!st->uploaded) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_state: dxo_vm_state_upload has not been called");
    const int d = st->d;
    const int64_t n = st->n;
    if (n == 0) return DXO_OK;
    if (!deps || !C_tang) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_state: NULL array");
    if (((uintptr_t)deps | (uintptr_t)C_tang | (uintptr_t)sigma | (uintptr_t)dp) & 7u)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_state: arrays must be 8-byte aligned");
    VmLaunch L{make_const(*prm), d};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    st->has_result = false;
    const size_t sd = sizeof(double);
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = vm_launch(ctx, L, n, deps, st->sigma_n, st->p, C_tang, st->sigma, st->dp, s);
        if (rc != DXO_OK) return rc;
        if (sigma) DXO_HIP(ctx, hipMemcpyAsync(sigma, st->sigma, (size_t)n * d * sd, hipMemcpyDeviceToDevice, s));
        if (dp) DXO_HIP(ctx, hipMemcpyAsync(dp, st->dp, (size_t)n * sd, hipMemcpyDeviceToDevice, s));
        rc = dxo_device_end(ctx, s);
        if (rc == DXO_OK) st->has_result = true;
        return rc;
    }
    if (!sigma || !dp) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_state: host sigma / dp are required");
    std::vector<dxo_span> in = {{deps, nullptr, d * sd}, {nullptr, nullptr, d * sd, st->sigma_n}, {nullptr, nullptr, sd, st->p}};
    int rc;
    if (ctx->vm_host_tangent && n >= ctx->vm_rebuild_min_points) {
        L.h_sigma = sigma;
        L.h_dp = dp;
        L.h_C_tang = C_tang;
        L.c.mark_indeterminate = 1;   // the mark (dp = -0.0) stays in the mirror's dp: p + (-0.0) == p at the commit
        std::vector<dxo_span> out = {{nullptr, nullptr, 0}, {nullptr, sigma, d * sd, st->sigma}, {nullptr, dp, sd, st->dp}};
        const int64_t saved_chunk = ctx->host_chunk_points;
        if (ctx->host_chunk_points > ctx->vm_rebuild_chunk_points) ctx->host_chunk_points = ctx->vm_rebuild_chunk_points;
        rc = dxo_run_host_pipeline(ctx, n, in, out, vm_chunk, &L, 1, vm_host_rebuild, true);
        ctx->host_chunk_points = saved_chunk;
    } else {
        std::vector<dxo_span> out = {{nullptr, C_tang, d * d * sd}, {nullptr, sigma, d * sd, st->sigma}, {nullptr, dp, sd, st->dp}};
        rc = dxo_run_host_pipeline(ctx, n, in, out, vm_chunk, &L, 1, nullptr, true);
    }
    if (rc == DXO_OK) st->has_result = true;
    return rc;
}

// ------------------------------------------------------------------ history update
// End of a load step in the reference: `p.x.petsc_vec.axpy(1.0, dp.x.petsc_vec)` and
// `sigma_n.x.array[:] = sigma.ref_coefficient.x.array` (demo_plasticity_von_mises.py:564-565). With the state
// resident on the device this is one fused lane-linear pass instead of two host loops plus two uploads.
#ifndef DXO_COMMIT_X2
#define DXO_COMMIT_X2 1
#endif
#ifndef DXO_COMMIT_BLOCKS_PER_CU
#define DXO_COMMIT_BLOCKS_PER_CU 256   // 10^7 points, d = 6 (1.2 GB per call), scripts/exp/commit_ab.py: 16 workgroups per CU 0.215-0.237 ms, 64 0.202-0.223, 128 0.212,
                                       // 256 0.198-0.203, 512 0.197-0.204 (8-byte accesses at 256: 0.201) — the grid, not the access width
#endif
namespace {
__global__ __launch_bounds__(DXO_BLOCK) void vm_commit(int64_t n_p, int64_t n_s, double* __restrict__ p,
                                                       const double* __restrict__ dp, double* __restrict__ sigma_n,
                                                       const double* __restrict__ sigma) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < n_p; i += stride) p[i] += dp[i];
    for (int64_t i = i0; i < n_s; i += stride) sigma_n[i] = sigma[i];
}
// the same pass in 16-byte accesses (arrays 16-byte aligned, which every allocation of the library and of torch is): n_p2 / n_s2 pairs, then the
// odd tail of p. Same additions, same bits.
__global__ __launch_bounds__(DXO_BLOCK) void vm_commit_x2(int64_t n_p, int64_t n_s2, dxo_f64x2* __restrict__ p2, const dxo_f64x2* __restrict__ dp2,
                                                          dxo_f64x2* __restrict__ sigma_n2, const dxo_f64x2* __restrict__ sigma2) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_p2 = n_p >> 1;
    for (int64_t i = i0; i < n_s2; i += stride) sigma_n2[i] = sigma2[i];
    for (int64_t i = i0; i < n_p2; i += stride) {
        dxo_f64x2 a = p2[i];
        const dxo_f64x2 b = dp2[i];
        a.x += b.x;
        a.y += b.y;
        p2[i] = a;
    }
    if ((n_p & 1) && i0 == 0) reinterpret_cast<double*>(p2)[n_p - 1] += reinterpret_cast<const double*>(dp2)[n_p - 1];
}
}  // namespace

extern "C" int dxo_vm_commit_state(dxo_ctx* ctx, int d, int64_t n, double* p, const double* dp, double* sigma_n,
                                   const double* sigma) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (d != 4 && d != 6) return dxo_fail(ctx, DXO_E_DIM, "dxo_vm_commit_state: d must be 4 or 6");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_commit_state: n < 0");
    if (n > 0 && (!p || !dp || !sigma_n || !sigma)) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_commit_state: NULL array");
    if (((uintptr_t)p | (uintptr_t)dp | (uintptr_t)sigma_n | (uintptr_t)sigma) & 7u)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_vm_commit_state: arrays must be 8-byte aligned");
    if (n == 0) return DXO_OK;
    hipStream_t s = dxo_launch_stream(ctx);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    const bool x2 = ((((uintptr_t)p | (uintptr_t)dp | (uintptr_t)sigma_n | (uintptr_t)sigma) & 15u) == 0) && DXO_COMMIT_X2;   // n * d is even (d = 4, 6)
    int64_t blocks = ((x2 ? n * d / 2 : n * d) + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * DXO_COMMIT_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    if (x2) hipLaunchKernelGGL(vm_commit_x2, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, n, n * d / 2, reinterpret_cast<dxo_f64x2*>(p), reinterpret_cast<const dxo_f64x2*>(dp),
                               reinterpret_cast<dxo_f64x2*>(sigma_n), reinterpret_cast<const dxo_f64x2*>(sigma));
    else hipLaunchKernelGGL(vm_commit, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, n, n * d, p, dp, sigma_n, sigma);
    return dxo_device_end(ctx, s);
}

// -0.0 -> +0.0 in dp: the producer's mark for the reference's 0/0 point (option "vm_mark_indeterminate") is consumed by
// dxo_vm_expand_tangent; afterwards dp is the reference's +0 again. Reads 8 B/point, writes only marked points.
__global__ __launch_bounds__(DXO_BLOCK) void vm_clear_marks(int64_t n, double* __restrict__ dp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = dp[i];
        if (v == 0.0 && __builtin_signbit(v)) dp[i] = 0.0;
    }
}

extern "C" int dxo_vm_clear_marks(dxo_ctx* ctx, int64_t n, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_clear_marks: n < 0");
    if (n == 0) return DXO_OK;
    if (!dp) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_clear_marks: dp is NULL");
    if ((uintptr_t)dp & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_vm_clear_marks: dp must be 8-byte aligned");
    hipStream_t s = dxo_launch_stream(ctx);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    int64_t blocks = (n + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(vm_clear_marks, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, n, dp);
    return dxo_device_end(ctx, s);
}

extern "C" int dxo_vm_state_commit(dxo_ctx* ctx, dxo_vm_state* st) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_vm_state_commit: state is NULL");
    if (st->n == 0) return DXO_OK;   // an empty partition (a rank without cells) has nothing to update: not an error
    if (!st->has_result)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_vm_state_commit: no call since the last upload / commit — nothing to commit");
    int rc = dxo_vm_commit_state(ctx, st->d, st->n, st->p, st->dp, st->sigma_n, st->sigma);
    if (rc != DXO_OK) return rc;
    DXO_HIP(ctx, hipStreamSynchronize(dxo_launch_stream(ctx)));   // the pipeline's streams read the mirror next
    st->has_result = false;
    return DXO_OK;
}
