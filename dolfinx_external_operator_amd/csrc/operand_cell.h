// operand_cell.h — lane = cell form of the strain evaluation for the standard elements (operand.hip, vm_field.hip).
//
// The wave-group kernels of operand_core.h (lane = (cell, quadrature point), dofs and tables in LDS) are LDS-bound on
// Q2 hexahedra: every fp64 FMA of Gref = sum_a u_a (x) dphi_a(q) reads 2/3 of an operand from LDS (~210 eight-byte LDS
// reads per point, 76 % LDS busy). Here ONE LANE OWNS ONE CELL: the cell's dof values and vertex
// coordinates are gathered once into registers (81 + 24 doubles for Q2 / Q1-geometry hexahedra) and serve all NQ points
// of the cell; the reference tables are wave-uniform, so they arrive as scalar operands (s_load -> SGPR sources of the
// FMAs) and the contraction touches no LDS at all. The strains of the wave's 64 cells x NQ points are parked in the
// wave's LDS slice E (point-major, the order of the output arrays) and then
//   * operand_cell_eps  writes them out with lane-linear stores (dxo_eval_operand, kind EPS_MANDEL), or
//   * vm_field_cell     runs vm_tile's body on them, 64 consecutive points at a time (dxo_von_mises_field): the return
//                       map reads its strain increment from E instead of from memory; sigma_n / sigma / C_tang move
//                       exactly as in vm_tile.
// Everything is unrolled at compile time, hence one instantiation per (gdim, nodes, points, vertices per cell); other
// elements, entity lists and other operand kinds keep the wave-group kernels.
// Reference: fem.Expression(eps(Du), points).eval inside evaluate_operands
// (src/dolfinx_external_operator/external_operator.py:386-402), demo_plasticity_von_mises.py:225-227, 445-456.
#pragma once

#include "dxo_common.h"
#include "operand_core.h"
#include "vm_core.h"

namespace {

template <int G, int ND, int NQ, int NG>
struct CellGeom {
    static constexpr int D = G == 2 ? 4 : 6;
    static constexpr int PTS = DXO_WAVE * NQ;                 // points of one wave-group (64 cells)
    static constexpr int ES = (NQ * D) | 1;                   // odd per-cell stride of E: lanes hit different banks
    static constexpr int E_DOUBLES = DXO_WAVE * ES + 1;
};

// dofs and vertices of `cell` into registers
template <int G, int ND, int NG>
__device__ __forceinline__ void cell_gather(const OperandDev& m, int64_t cell, const double* __restrict__ u,
                                            double (&X)[NG][G], double (&U)[ND][G]) {
    int32_t gn[NG], un[ND];
#pragma unroll
    for (int v = 0; v < NG; ++v) gn[v] = m.geom_dofmap[cell * NG + v];
#pragma unroll
    for (int a = 0; a < ND; ++a) un[a] = m.dofmap[cell * ND + a];
#pragma unroll
    for (int v = 0; v < NG; ++v)
#pragma unroll
        for (int j = 0; j < G; ++j) X[v][j] = m.x[(int64_t)gn[v] * G + j];
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int i = 0; i < G; ++i) U[a][i] = u[(int64_t)un[a] * G + i];
}

// Mandel strain of point Q of the cell held in (X, U); tables through scalar loads (uniform addresses)
template <int G, int ND, int NQ, int NG, int Q>
__device__ __forceinline__ void cell_point_eps(const double* __restrict__ dphi, const double* __restrict__ dpsi,
                                               const double (&X)[NG][G], const double (&U)[ND][G],
                                               double (&e)[CellGeom<G, ND, NQ, NG>::D]) {
    double J[G][G], K[G][G];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int k = 0; k < G; ++k) J[j][k] = 0.0;
#pragma unroll
    for (int v = 0; v < NG; ++v)
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < G; ++k) J[j][k] += X[v][j] * dpsi[(Q * NG + v) * G + k];
    (void)invert<G>(J, K);
    double gref[G][G];
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int k = 0; k < G; ++k) gref[i][k] = 0.0;
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const double dk = dphi[(Q * ND + a) * G + k];
#pragma unroll
            for (int i = 0; i < G; ++i) gref[i][k] += U[a][i] * dk;
        }
    double g[G][G];
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
        for (int j = 0; j < G; ++j) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < G; ++k) s += gref[i][k] * K[k][j];
            g[i][j] = s;
        }
    double val[G];
#pragma unroll
    for (int i = 0; i < G; ++i) val[i] = 0.0;
    shape_operand<G, G, DXO_OPERAND_EPS_MANDEL>(val, g, e);
}

template <int G, int ND, int NQ, int NG, int Q>
__device__ __forceinline__ void cell_all_points(const double* __restrict__ dphi, const double* __restrict__ dpsi,
                                                const double (&X)[NG][G], const double (&U)[ND][G], double* Ec) {
    if constexpr (Q < NQ) {
        constexpr int D = CellGeom<G, ND, NQ, NG>::D;
        double e[D];
        cell_point_eps<G, ND, NQ, NG, Q>(dphi, dpsi, X, U, e);
#pragma unroll
        for (int k = 0; k < D; ++k) Ec[Q * D + k] = e[k];
        cell_all_points<G, ND, NQ, NG, Q + 1>(dphi, dpsi, X, U, Ec);
    }
}

// stage 1 of both kernels: strains of the wave's 64 cells -> E[cell_in_group * ES + q * D + k]
template <int G, int ND, int NQ, int NG>
__device__ __forceinline__ void cell_stage_strain(const OperandDev& m, int64_t first_cell, int64_t last_valid_cell, int lane,
                                                  const double* __restrict__ u, double* E) {
    using CG = CellGeom<G, ND, NQ, NG>;
    int64_t cell = first_cell + lane;
    if (cell > last_valid_cell) cell = last_valid_cell;       // idle lanes recompute the last cell (never stored)
    double X[NG][G], U[ND][G];
    cell_gather<G, ND, NG>(m, cell, u, X, U);
    cell_all_points<G, ND, NQ, NG, 0>(m.dphi, m.dpsi, X, U, E + lane * CG::ES);
}

// ------------------------------------------------------------------ dxo_eval_operand, kind EPS_MANDEL, all cells
template <int G, int ND, int NQ, int NG>
__global__ __launch_bounds__(DXO_BLOCK, 1) void operand_cell_eps(OperandDev m, const double* __restrict__ u, int64_t n_cells,
                                                                 double* __restrict__ out) {
    using CG = CellGeom<G, ND, NQ, NG>;
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ double lds[WAVES * CG::E_DOUBLES];
    const int lane = threadIdx.x & (DXO_WAVE - 1), wave = threadIdx.x >> 6;
    double* E = lds + wave * CG::E_DOUBLES;
    const int64_t n_groups = (n_cells + DXO_WAVE - 1) / DXO_WAVE;
    const GroupWalk walk = xcd_group_walk(n_groups, WAVES, wave);
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * DXO_WAVE;
        const int ncell = n_cells - c0 < DXO_WAVE ? (int)(n_cells - c0) : DXO_WAVE;
        cell_stage_strain<G, ND, NQ, NG>(m, c0, n_cells - 1, lane, u, E);
        op_fence();
        const int nval = ncell * NQ * CG::D;
        double* g_o = out + c0 * (NQ * CG::D);
        for (int idx = lane; idx < nval; idx += DXO_WAVE) {
            const int c = idx / (NQ * CG::D), r = idx - c * (NQ * CG::D);
            __builtin_nontemporal_store(E[c * CG::ES + r], g_o + idx);
        }
        op_fence();
    }
}

// ------------------------------------------------------------------ dxo_von_mises_field
template <int G, int ND, int NQ, int NG, bool NT>
__global__ __launch_bounds__(DXO_BLOCK, 1) void vm_field_cell(VmConst c, OperandDev m, int64_t cell0, int64_t n_cells,
                                                              const double* __restrict__ u, const double* __restrict__ sigma_n,
                                                              const double* __restrict__ p, double* __restrict__ C_tang,
                                                              double* __restrict__ sigma, double* __restrict__ dp_out) {
    using CG = CellGeom<G, ND, NQ, NG>;
    constexpr int D = CG::D;
    using T = VmTile<D>;
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    constexpr int WD = (CG::E_DOUBLES + 1 + T::X_DOUBLES + T::Y_DOUBLES + 1) & ~1;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * WD];
    const int lane = threadIdx.x & (DXO_WAVE - 1), wave = threadIdx.x >> 6;
    double* X = lds + wave * WD;                       // 16-byte aligned: WD is even
    double* Y = X + T::X_DOUBLES;
    double* E = Y + T::Y_DOUBLES;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
    const int64_t n_groups = (n_cells + DXO_WAVE - 1) / DXO_WAVE;
    const GroupWalk walk = xcd_group_walk(n_groups, WAVES, wave);
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * DXO_WAVE;                       // first cell of the group, relative to cell0
        const int ncell = n_cells - c0 < DXO_WAVE ? (int)(n_cells - c0) : DXO_WAVE;
        cell_stage_strain<G, ND, NQ, NG>(m, cell0 + c0, cell0 + n_cells - 1, lane, u, E);
        wave_lds_fence();
        const int npts_group = ncell * NQ;
        // the group's points, 64 at a time, through vm_tile's body
#pragma unroll 1
        for (int t = 0; t < NQ; ++t) {
            const int P0 = t * DXO_WAVE;
            if (P0 >= npts_group) break;
            const int npts = npts_group - P0 < DXO_WAVE ? npts_group - P0 : DXO_WAVE;
            const int64_t p0 = c0 * NQ + P0;                     // relative to the state / output arrays passed in
            const int nvec = npts * T::CH_VEC;
            const dxo_f64x2* g_s = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                const int idx = k * DXO_WAVE + lane;
                Y2[idx] = idx < nvec ? g_s[idx] : dxo_f64x2{0.0, 0.0};
            }
            const double p_l = lane < npts ? p[p0 + lane] : 0.0;
            const int P = P0 + lane, pc = P / NQ, pq = P - pc * NQ;
            double e[D];
#pragma unroll
            for (int k = 0; k < D; ++k) e[k] = lane < npts ? E[pc * CG::ES + pq * D + k] : 0.0;
            wave_lds_fence();
            double sn[D];
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                const dxo_f64x2 b2 = Y2[lane * T::CH_VEC + k];
                sn[2 * k] = b2.x;
                sn[2 * k + 1] = b2.y;
            }
            wave_lds_fence();
            double sig[D], nrm[D], dp, a, b;
            vm_return_map<D>(c, e, sn, p_l, sig, dp, nrm, a, b);
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                X2[lane * T::CH_VEC + k] = dxo_f64x2{sig[2 * k], sig[2 * k + 1]};
                Y2[lane * (T::ST / 2) + k] = dxo_f64x2{nrm[2 * k], nrm[2 * k + 1]};
            }
            Y2[lane * (T::ST / 2) + T::CH_VEC] = dxo_f64x2{a, b};
            wave_lds_fence();
            if (lane < npts) store8<NT>(dp_out + p0 + lane, dp);
            dxo_f64x2* g_o = reinterpret_cast<dxo_f64x2*>(sigma + p0 * D);
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                const int idx = k * DXO_WAVE + lane;
                if (idx < nvec) store16<NT>(g_o + idx, X2[idx]);
            }
            vm_store_tangent<D, NT>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
            wave_lds_fence();
        }
    }
}

// Elements with a lane = cell instantiation (the same list as adjoint_cell.h)
#define DXO_OPERAND_CELL_CASES(X_)                                   \
    X_(3, 27, 8, 8)  /* Q2 hexahedra, 2x2x2 Gauss (BASELINE config 2/3) */ \
    X_(3, 8, 8, 8)   /* Q1 hexahedra */                              \
    X_(3, 10, 4, 4)  /* P2 tetrahedra */                             \
    X_(2, 6, 3, 3)   /* P2 triangles, 3-point rule (the reference demos) */ \
    X_(2, 9, 4, 4)   /* Q2 quadrilaterals */

inline int cell_grid(const dxo_ctx* ctx, int64_t n_cells) {
    const int64_t n_groups = (n_cells + DXO_WAVE - 1) / DXO_WAVE;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 4;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;      // whole rounds over the 8 XCDs (xcd_group_walk)
    return (int)blocks;
}

// true if a specialised kernel was launched
inline bool launch_operand_cell_eps(const dxo_ctx* ctx, const dxo_mesh* mesh, const double* u, int64_t n_cells, double* out,
                                    hipStream_t s) {
    const OperandDev& v = mesh->dev;
    const int grid = cell_grid(ctx, n_cells);
#define DXO_CASE(G_, ND_, NQ_, NG_)                                                                                      \
    if (mesh->gdim == G_ && v.ndofs == ND_ && v.nq == NQ_ && v.ngeom == NG_) {                                            \
        hipLaunchKernelGGL((operand_cell_eps<G_, ND_, NQ_, NG_>), dim3(grid), dim3(DXO_BLOCK), 0, s, v, u, n_cells, out); \
        return true;                                                                                                      \
    }
    DXO_OPERAND_CELL_CASES(DXO_CASE)
#undef DXO_CASE
    return false;
}

inline bool launch_vm_field_cell(const dxo_ctx* ctx, const VmConst& c, const dxo_mesh* mesh, int64_t cell0, int64_t n_cells,
                                 const double* u, const double* sigma_n, const double* p, double* C_tang, double* sigma,
                                 double* dp, hipStream_t s) {
    const OperandDev& v = mesh->dev;
    const int grid = cell_grid(ctx, n_cells);
    const bool nt = ctx->nontemporal != 0;
#define DXO_CASE(G_, ND_, NQ_, NG_)                                                                                       \
    if (mesh->gdim == G_ && v.ndofs == ND_ && v.nq == NQ_ && v.ngeom == NG_) {                                             \
        if (nt) hipLaunchKernelGGL((vm_field_cell<G_, ND_, NQ_, NG_, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, c, v, cell0, n_cells, u, sigma_n, p, C_tang, sigma, dp); \
        else    hipLaunchKernelGGL((vm_field_cell<G_, ND_, NQ_, NG_, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, c, v, cell0, n_cells, u, sigma_n, p, C_tang, sigma, dp); \
        return true;                                                                                                       \
    }
    DXO_OPERAND_CELL_CASES(DXO_CASE)
#undef DXO_CASE
    return false;
}

}  // namespace
