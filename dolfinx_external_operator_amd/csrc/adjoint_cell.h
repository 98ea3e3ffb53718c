// adjoint_cell.h — lane = cell form of the virtual-work scatter for the standard elements (included by adjoint.hip).
//
// One lane owns one cell: the cell's vertex coordinates and the pulled-back dual tensors of ALL its quadrature points live
// in registers, the reference tables are wave-uniform and arrive as scalar operands (s_load -> SGPR sources of the fp64
// FMAs), so the kernel has no LDS traffic at all; with the element vectors laid out fe[local node][cell][component] the 64
// lanes of a wave store 64 consecutive entries per (node, component). Everything is unrolled at compile time, hence one
// instantiation per (gdim, nodes per cell, points per cell, vertices per cell).
#pragma once

#include "dxo_common.h"
#include "operand_core.h"

namespace {

template <int G, int ND, int NQ, int NG>
__global__ __launch_bounds__(DXO_BLOCK) void adjoint_cell_eps(const double* __restrict__ dphi, const double* __restrict__ dpsi,
                                                              const double* __restrict__ wq, const double* __restrict__ x,
                                                              const int32_t* __restrict__ geom_dofmap,
                                                              const double* __restrict__ S, int64_t n_cells,
                                                              double* __restrict__ fe) {
    constexpr int D = G == 2 ? 4 : 6;
    constexpr double r2 = 0.70710678118654752440;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < n_cells; cell += stride) {
        double X[NG][G];
#pragma unroll
        for (int v = 0; v < NG; ++v) {
            const int64_t node = geom_dofmap[cell * NG + v];
#pragma unroll
            for (int j = 0; j < G; ++j) X[v][j] = x[node * G + j];
        }
        double T[NQ][G][G];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
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
                    for (int k = 0; k < G; ++k) J[j][k] += X[v][j] * dpsi[(q * NG + v) * G + k];
            const double det = invert<G>(J, K);
            const double scale = wq[q] * fabs(det);
            const double* Sp = S + (cell * NQ + q) * D;
            double s[D];
#pragma unroll
            for (int k = 0; k < D; ++k) s[k] = Sp[k];
            double gh[G][G];
            if constexpr (G == 2) {
                gh[0][0] = s[0]; gh[1][1] = s[1]; gh[0][1] = gh[1][0] = r2 * s[3];
            } else {
                gh[0][0] = s[0]; gh[1][1] = s[1]; gh[2][2] = s[2];
                gh[0][1] = gh[1][0] = r2 * s[3]; gh[0][2] = gh[2][0] = r2 * s[4]; gh[1][2] = gh[2][1] = r2 * s[5];
            }
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    double t = 0.0;
#pragma unroll
                    for (int j = 0; j < G; ++j) t += gh[i][j] * K[k][j];
                    T[q][i][k] = scale * t;
                }
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int k = 0; k < G; ++k) acc += T[q][i][k] * dphi[(q * ND + a) * G + k];
                fe[((int64_t)a * n_cells + cell) * G + i] = acc;
            }
        }
    }
}

#ifdef DXO_EXPERIMENTS      // measured, not shipped (profiles/r04_adjoint_experiments.txt)
// lane = cell form of the matrix-free tangent action K v (dxo_tangent_apply) for small simplicial elements: the cell's dof values
// of v, its vertices and the pulled-back tensors of all its points in registers, tables as scalar operands, no LDS; the lane reads
// the D*D tangent entries of each of its points as 16-byte pieces of its own contiguous row block. On P2 triangles (the
// reference demos' element: 6 nodes, 3 points, demo_plasticity_von_mises.py:230,245,295) the wave-group kernel spends its time in
// LDS traffic for 3-point cells that fill only 63 lanes; this form moves the same bytes with a fraction of the instructions.
template <int G, int ND, int NQ, int NG>
__global__ __launch_bounds__(DXO_BLOCK) void tangent_cell(const double* __restrict__ dphi, const double* __restrict__ dpsi,
                                                          const double* __restrict__ wq, const double* __restrict__ x,
                                                          const int32_t* __restrict__ geom_dofmap, const int32_t* __restrict__ dofmap,
                                                          const double* __restrict__ C_tang, const double* __restrict__ v,
                                                          int64_t n_cells, double* __restrict__ fe) {
    constexpr int D = G == 2 ? 4 : 6;
    constexpr double r2 = 0.70710678118654752440;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < n_cells; cell += stride) {
        double X[NG][G], V[ND][G];
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            const int64_t node = dofmap[cell * ND + a];
#pragma unroll
            for (int i = 0; i < G; ++i) V[a][i] = v[node * G + i];
        }
#pragma unroll
        for (int vv = 0; vv < NG; ++vv) {
            const int64_t node = geom_dofmap[cell * NG + vv];
#pragma unroll
            for (int j = 0; j < G; ++j) X[vv][j] = x[node * G + j];
        }
        double T[NQ][G][G];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const dxo_f64x2* Cp = reinterpret_cast<const dxo_f64x2*>(C_tang + (cell * NQ + q) * (D * D));
            dxo_f64x2 Cq[D * D / 2];
#pragma unroll
            for (int k = 0; k < D * D / 2; ++k) Cq[k] = Cp[k];
            double J[G][G], K[G][G];
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] = 0.0;
#pragma unroll
            for (int vv = 0; vv < NG; ++vv)
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int k = 0; k < G; ++k) J[j][k] += X[vv][j] * dpsi[(q * NG + vv) * G + k];
            const double det = invert<G>(J, K);
            const double scale = wq[q] * fabs(det);
            double gref[G][G];
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < G; ++k) gref[i][k] = 0.0;
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int k = 0; k < G; ++k) gref[i][k] += V[a][i] * dphi[(q * ND + a) * G + k];
            double g[G][G];
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < G; ++k) sacc += gref[i][k] * K[k][j];
                    g[i][j] = sacc;
                }
            double e[D], val[G];
#pragma unroll
            for (int i = 0; i < G; ++i) val[i] = 0.0;
            shape_operand<G, G, DXO_OPERAND_EPS_MANDEL>(val, g, e);
            double t[D];
#pragma unroll
            for (int r = 0; r < D; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int cc = 0; cc < D; cc += 2) {
                    const dxo_f64x2 c2 = Cq[(r * D + cc) / 2];
                    acc += c2.x * e[cc];
                    acc += c2.y * e[cc + 1];
                }
                t[r] = acc;
            }
            double gh[G][G];
            if constexpr (G == 2) {
                gh[0][0] = t[0]; gh[1][1] = t[1]; gh[0][1] = gh[1][0] = r2 * t[3];
            } else {
                gh[0][0] = t[0]; gh[1][1] = t[1]; gh[2][2] = t[2];
                gh[0][1] = gh[1][0] = r2 * t[3]; gh[0][2] = gh[2][0] = r2 * t[4]; gh[1][2] = gh[2][1] = r2 * t[5];
            }
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < G; ++j) tt += gh[i][j] * K[k][j];
                    T[q][i][k] = scale * tt;
                }
        }
#pragma unroll
        for (int a = 0; a < ND; ++a) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int k = 0; k < G; ++k) acc += T[q][i][k] * dphi[(q * ND + a) * G + k];
                fe[((int64_t)a * n_cells + cell) * G + i] = acc;
            }
        }
    }
}
#endif

#ifdef DXO_EXPERIMENTS
inline bool launch_tangent_cell(const dxo_ctx* ctx, const dxo_mesh* m, const double* C_tang, const double* v, double* fe, hipStream_t s) {
    const OperandDev& d = m->dev;
    int64_t blocks = (m->num_cells + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (m->gdim == 2 && d.ndofs == 6 && d.nq == 3 && d.ngeom == 3) {      // P2 triangles, 3-point rule (the reference demos)
        hipLaunchKernelGGL((tangent_cell<2, 6, 3, 3>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, d.dphi, d.dpsi, m->d_wq, d.x, d.geom_dofmap,
                           d.dofmap, C_tang, v, m->num_cells, fe);
        return true;
    }
    return false;
}
#endif

// launches the specialised kernel if the mesh's element is one of the instantiated ones; returns false otherwise
#ifndef DXO_AC_BLOCKS_PER_CU
#define DXO_AC_BLOCKS_PER_CU 128   // adjoint_cell_eps grid; 1 / 8 / 32 / 128 workgroups per CU on P2 triangles: 0.302 / 0.306 / 0.293 / 0.290 ms per call
#endif
inline bool launch_adjoint_cell_eps(const dxo_ctx* ctx, const dxo_mesh* m, const double* S, double* fe, hipStream_t s) {
    const OperandDev& v = m->dev;
    int64_t blocks = (m->num_cells + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * DXO_AC_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
#define DXO_CELL_CASE(G_, ND_, NQ_, NG_)                                                                                 \
    if (m->gdim == G_ && v.ndofs == ND_ && v.nq == NQ_ && v.ngeom == NG_) {                                               \
        hipLaunchKernelGGL((adjoint_cell_eps<G_, ND_, NQ_, NG_>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, v.dphi, v.dpsi, \
                           m->d_wq, v.x, v.geom_dofmap, S, m->num_cells, fe);                                            \
        return true;                                                                                                      \
    }
    DXO_CELL_CASE(3, 27, 8, 8)    // Q2 hexahedra, 2x2x2 Gauss
    DXO_CELL_CASE(3, 8, 8, 8)     // Q1 hexahedra
    DXO_CELL_CASE(3, 10, 4, 4)    // P2 tetrahedra
    DXO_CELL_CASE(2, 6, 3, 3)     // P2 triangles, 3-point rule (the reference demos)
    DXO_CELL_CASE(2, 9, 4, 4)     // Q2 quadrilaterals
#undef DXO_CELL_CASE
    return false;
}

}  // namespace
