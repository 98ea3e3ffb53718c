// operand_cell.h — lane = cell form of the strain evaluation for the 2-D standard elements (operand.hip).
//
// The wave-group kernel of operand_core.h (lane = (cell, quadrature point), dofs and tables in LDS) pays one LDS read
// per operand of every FMA. Here ONE LANE OWNS ONE CELL: the cell's dof values and vertex coordinates are gathered once
// into registers and serve all NQ points of the cell; the reference tables are wave-uniform, so they arrive as scalar
// operands (s_load -> SGPR sources of the FMAs) and the contraction touches no LDS at all. The strains of the wave's
// 64 cells x NQ points are parked in the wave's LDS slice E (point-major, the order of the output array) and written
// out with lane-linear stores. Measured (P2 triangles, 6*10^6 points, dxo_eval_operand kind EPS_MANDEL): 0.053 ms =
// 6.5 TB/s against 0.072 ms for the wave-group kernel.
// Where it does NOT pay (measured on MI355X, round 2, and therefore not instantiated): Q2 hexahedra — 81 + 24 doubles
// of state per lane mean 256 VGPRs + 133 AGPRs, one wave per SIMD, and neither the gather latency nor the ~2500 FMAs
// per cell are hidden: 0.50 ms against 0.46 ms (wave-group); a fused form that ran vm_tile's body on E (eight tiles of
// 64 points per wave-group, 129 KB of LDS per workgroup, one wave per SIMD) took 1.64 ms against 1.00 ms for
// vm_field<3>, and 0.37 against 0.30 ms on P2 triangles, so dxo_von_mises_field keeps the wave-group kernel.
// Everything is unrolled at compile time, hence one instantiation per (gdim, nodes, points, vertices per cell); other
// elements, entity lists and other operand kinds keep the wave-group kernels.
// Reference: fem.Expression(eps(Du), points).eval inside evaluate_operands
// (src/dolfinx_external_operator/external_operator.py:386-402), demo_plasticity_von_mises.py:225-227.
#pragma once

#include "dxo_common.h"
#include "operand_core.h"

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

// strains of the wave's 64 cells -> E[cell_in_group * ES + q * D + k]
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
__global__ __launch_bounds__(DXO_BLOCK) void operand_cell_eps(OperandDev m, const double* __restrict__ u, int64_t n_cells,
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

// Elements with a lane = cell instantiation
#define DXO_OPERAND_CELL_CASES(X_)                                   \
    X_(2, 6, 3, 3)   /* P2 triangles, 3-point rule (the reference demos) */ \
    X_(2, 9, 4, 4)   /* Q2 quadrilaterals */

inline int cell_grid(const dxo_ctx* ctx, int64_t n_cells) {
    const int64_t n_groups = (n_cells + DXO_WAVE - 1) / DXO_WAVE;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
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

}  // namespace
