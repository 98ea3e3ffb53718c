// heat.hip — nonlinear heat flux q(T, sigma) = -k(T) sigma, k = 1/(A + B T), with both partial
// derivatives, one fused launch.
//
// Algorithm: doc/demo/demo_nonlinear_heat_equation_part2.py:215-261 (k :215-216, q_impl :219-230,
// dqdT_impl :243-247, dqdsigma_impl :259-261); the same three kernels are the reference's own
// unit-test kernels, test/test_external_operators_evaluation.py:69-86.
//
// Roofline: HBM, 88 B per point at gdim = 2 when all three outputs are requested
// (read 1+2, write 2+2+4 doubles). gdim = 2 path: every access is lane-linear; the 2x2 block of
// d q/d sigma is written in output order (lane -> one 16-byte row) with k fetched from the owning
// lane by a wave shuffle, so no store instruction is strided.
#include "dxo_common.h"

namespace {

template <int G>
__global__ __launch_bounds__(DXO_BLOCK) void heat_point(double A, double B, int64_t n, const double* __restrict__ T,
                                                        const double* __restrict__ sigma, double* __restrict__ q,
                                                        double* __restrict__ dqdT, double* __restrict__ dqds) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double k = 1.0 / (A + B * T[i]);   // :216
        const double mk = -k;
        const double bk2 = B * (k * k);          // B * k**2 (:246)
#pragma unroll
        for (int a = 0; a < G; ++a) {
            const double s = sigma[i * G + a];
            if (q) q[i * G + a] = mk * s;        // :228
            if (dqdT) dqdT[i * G + a] = bk2 * s; // :246
            if (dqds) {
#pragma unroll
                for (int b = 0; b < G; ++b) dqds[(i * G + a) * G + b] = mk * (a == b ? 1.0 : 0.0);  // :260
            }
        }
    }
}

// gdim == 2, 16-byte aligned arrays: lane-linear 16-byte traffic everywhere.
__global__ __launch_bounds__(DXO_BLOCK) void heat_g2(double A, double B, int64_t n, const double* __restrict__ T,
                                                     const dxo_f64x2* __restrict__ sigma, dxo_f64x2* __restrict__ q,
                                                     dxo_f64x2* __restrict__ dqdT, dxo_f64x2* __restrict__ dqds) {
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int64_t n_round = (n + DXO_WAVE - 1) / DXO_WAVE * DXO_WAVE;  // whole waves stay convergent for the shuffles
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool ok = i < n;
        const double t = ok ? T[i] : 0.0;
        const dxo_f64x2 s = ok ? sigma[i] : dxo_f64x2{0.0, 0.0};
        const double k = 1.0 / (A + B * t);
        const double mk = -k;
        const double bk2 = B * (k * k);
        if (ok && q) __builtin_nontemporal_store(dxo_f64x2{mk * s.x, mk * s.y}, q + i);
        if (ok && dqdT) __builtin_nontemporal_store(dxo_f64x2{bk2 * s.x, bk2 * s.y}, dqdT + i);
        if (dqds) {
            const int64_t wave_p0 = i - lane;  // first point of this wave's 64-point tile
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = it * DXO_WAVE + lane;  // 16-byte row index inside the tile: point c/2, row c%2
                const double mk_src = __shfl(mk, c >> 1);
                const int row = c & 1;
                const dxo_f64x2 v = {mk_src * (row == 0 ? 1.0 : 0.0), mk_src * (row == 1 ? 1.0 : 0.0)};
                if (wave_p0 + (c >> 1) < n) __builtin_nontemporal_store(v, dqds + wave_p0 * 2 + c);
            }
        }
    }
}

struct HeatLaunch {
    double A, B;
    int gdim;
    bool want_q, want_dT, want_ds;
};

int heat_launch(dxo_ctx* ctx, const HeatLaunch& L, int64_t n, const double* T, const double* sigma, double* q,
                double* dqdT, double* dqds, hipStream_t s) {
    if (n == 0) return DXO_OK;
    const int grid = dxo_grid_for_tiles(ctx, (n + DXO_BLOCK - 1) / DXO_BLOCK, 1);
    const uintptr_t all = (uintptr_t)sigma | (uintptr_t)q | (uintptr_t)dqdT | (uintptr_t)dqds;
    if (L.gdim == 2 && (all & 15u) == 0) {
        hipLaunchKernelGGL(heat_g2, dim3(grid), dim3(DXO_BLOCK), 0, s, L.A, L.B, n, T, (const dxo_f64x2*)sigma,
                           (dxo_f64x2*)q, (dxo_f64x2*)dqdT, (dxo_f64x2*)dqds);
    } else if (L.gdim == 1) {
        hipLaunchKernelGGL((heat_point<1>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.A, L.B, n, T, sigma, q, dqdT, dqds);
    } else if (L.gdim == 2) {
        hipLaunchKernelGGL((heat_point<2>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.A, L.B, n, T, sigma, q, dqdT, dqds);
    } else {
        hipLaunchKernelGGL((heat_point<3>), dim3(grid), dim3(DXO_BLOCK), 0, s, L.A, L.B, n, T, sigma, q, dqdT, dqds);
    }
    return DXO_OK;
}

int heat_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const HeatLaunch& L = *static_cast<const HeatLaunch*>(user);
    int k = 0;
    double* q = L.want_q ? (double*)d_out[k++] : nullptr;
    double* dT = L.want_dT ? (double*)d_out[k++] : nullptr;
    double* ds = L.want_ds ? (double*)d_out[k++] : nullptr;
    return heat_launch(ctx, L, m, (const double*)d_in[0], (const double*)d_in[1], q, dT, ds, s);
}

}  // namespace

extern "C" int dxo_heat(dxo_ctx* ctx, double A, double B, int gdim, int64_t n, int mem, const double* T,
                        const double* sigma, double* q, double* dqdT, double* dqdsigma) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (gdim < 1 || gdim > 3) return dxo_fail(ctx, DXO_E_DIM, "dxo_heat: gdim must be 1, 2 or 3");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_heat: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_heat: bad mem");
    if (n > 0 && (!T || !sigma)) return dxo_fail(ctx, DXO_E_NULL, "dxo_heat: NULL input");
    const uintptr_t all = (uintptr_t)T | (uintptr_t)sigma | (uintptr_t)q | (uintptr_t)dqdT | (uintptr_t)dqdsigma;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_heat: arrays must be 8-byte aligned");
    HeatLaunch L{A, B, gdim, q != nullptr, dqdT != nullptr, dqdsigma != nullptr};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = heat_launch(ctx, L, n, T, sigma, q, dqdT, dqdsigma, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{T, nullptr, sd}, {sigma, nullptr, gdim * sd}};
    std::vector<dxo_span> out;
    if (q) out.push_back({nullptr, q, gdim * sd});
    if (dqdT) out.push_back({nullptr, dqdT, gdim * sd});
    if (dqdsigma) out.push_back({nullptr, dqdsigma, (size_t)gdim * gdim * sd});
    return dxo_run_host_pipeline(ctx, n, in, out, heat_chunk, &L, 1, nullptr, true);
}

// ------------------------------------------------------------------ scalar conductivity of the part-1 demo
// k(T) = 1 / (A + B T) and dk/dT = -B k^2: k_impl / dkdT_impl, demo_nonlinear_heat_equation_part1.py:251-271. The operator
// lives on a CG space there (its values go through the dofmap assigner, external_operator.py:286-287 = dxo_assign). One
// fused pass fills whichever of the two outputs are requested: 8 B read + 8 or 16 B written per value, HBM-bound.
namespace {

__global__ __launch_bounds__(DXO_BLOCK) void conductivity_kernel(double A, double B, int64_t n, const double* __restrict__ T,
                                                                 double* __restrict__ k_out, double* __restrict__ dk_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 2;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += stride) {
        if (i + 1 < n && (((uintptr_t)(T + i) | (uintptr_t)(k_out ? k_out + i : nullptr) | (uintptr_t)(dk_out ? dk_out + i : nullptr)) & 15u) == 0) {
            const dxo_f64x2 t = *reinterpret_cast<const dxo_f64x2*>(T + i);
            const dxo_f64x2 k = {1.0 / (A + B * t.x), 1.0 / (A + B * t.y)};                       // :254
            if (k_out) __builtin_nontemporal_store(k, reinterpret_cast<dxo_f64x2*>(k_out + i));
            if (dk_out) __builtin_nontemporal_store(dxo_f64x2{-B * (k.x * k.x), -B * (k.y * k.y)}, reinterpret_cast<dxo_f64x2*>(dk_out + i));   // :271
        } else {
            for (int64_t j = i; j < n && j < i + 2; ++j) {
                const double k = 1.0 / (A + B * T[j]);
                if (k_out) k_out[j] = k;
                if (dk_out) dk_out[j] = -B * (k * k);
            }
        }
    }
}

struct CondLaunch { double A, B; bool want_k, want_dk; };

int cond_launch(dxo_ctx* ctx, const CondLaunch& L, int64_t n, const double* T, double* k, double* dk, hipStream_t s) {
    if (n == 0) return DXO_OK;
    const int grid = dxo_grid_for_tiles(ctx, (n / 2 + DXO_BLOCK) / DXO_BLOCK, 1);
    hipLaunchKernelGGL(conductivity_kernel, dim3(grid), dim3(DXO_BLOCK), 0, s, L.A, L.B, n, T, k, dk);
    return DXO_OK;
}

int cond_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const CondLaunch& L = *static_cast<const CondLaunch*>(user);
    int o = 0;
    double* k = L.want_k ? (double*)d_out[o++] : nullptr;
    double* dk = L.want_dk ? (double*)d_out[o++] : nullptr;
    return cond_launch(ctx, L, m, (const double*)d_in[0], k, dk, s);
}

}  // namespace

extern "C" int dxo_conductivity(dxo_ctx* ctx, double A, double B, int64_t n, int mem, const double* T, double* k, double* dkdT) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_conductivity: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_conductivity: bad mem");
    if (n == 0 || (!k && !dkdT)) return DXO_OK;
    if (!T) return dxo_fail(ctx, DXO_E_NULL, "dxo_conductivity: T is NULL");
    if (((uintptr_t)T | (uintptr_t)k | (uintptr_t)dkdT) & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_conductivity: arrays must be 8-byte aligned");
    CondLaunch L{A, B, k != nullptr, dkdT != nullptr};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = cond_launch(ctx, L, n, T, k, dkdT, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{T, nullptr, sd}};
    std::vector<dxo_span> out;
    if (k) out.push_back({nullptr, k, sd});
    if (dkdT) out.push_back({nullptr, dkdT, sd});
    return dxo_run_host_pipeline(ctx, n, in, out, cond_chunk, &L, 1, nullptr, true);
}
