// operand_core.h — device code of the operand evaluation shared between translation units (operand.hip, vm_field.hip).
// Reference: evaluate_operands -> fem.Expression(operand, points).eval(mesh, entities),
// src/dolfinx_external_operator/external_operator.py:386-402.
#pragma once

#include "dxo_common.h"

// Device view of a mesh + element tables (plain struct: shared by every translation unit and by dxo_mesh below).
struct OperandDev {
    int nq, ndofs, ngeom;
    int cells_per_wave;            // floor(64 / nq)
    int wave_doubles;              // LDS doubles per wave
    int table_doubles;             // LDS doubles for the tables
    const double* phi;             // [nq][ndofs]
    const double* dphi;            // [nq][ndofs][G]
    const double* dpsi;            // [nq][ngeom][G]
    const int32_t* dofmap;         // [num_cells][ndofs]
    const int32_t* geom_dofmap;    // [num_cells][ngeom]
    const double* x;               // [num_geom_nodes][G]
};

namespace {


__device__ __forceinline__ void op_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G>
__device__ __forceinline__ void invert(const double (&J)[G][G], double (&K)[G][G]) {
    if constexpr (G == 2) {
        const double idet = 1.0 / (J[0][0] * J[1][1] - J[0][1] * J[1][0]);
        K[0][0] = J[1][1] * idet; K[0][1] = -J[0][1] * idet;
        K[1][0] = -J[1][0] * idet; K[1][1] = J[0][0] * idet;
    } else {
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double idet = 1.0 / (J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02);
        K[0][0] = c00 * idet; K[1][0] = c01 * idet; K[2][0] = c02 * idet;
        K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * idet;
        K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * idet;
        K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * idet;
        K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * idet;
        K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * idet;
        K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * idet;
    }
}

// Value size of the shaped operand.
template <int G, int BS, int KIND>
struct OperandShape {
    static constexpr int D = KIND == DXO_OPERAND_VALUE ? BS
                           : KIND == DXO_OPERAND_GRAD ? BS * G
                           : KIND == DXO_OPERAND_EPS_MANDEL ? (G == 2 ? 4 : 6)
                           : G * G;   // DXO_OPERAND_DEFGRAD
};

// grad u (BS x G, row = field component, column = direction) -> operand components
template <int G, int BS, int KIND>
__device__ __forceinline__ void shape_operand(const double (&val)[BS], const double (&g)[BS][G],
                                              double (&o)[OperandShape<G, BS, KIND>::D]) {
    constexpr double r2 = 0.70710678118654752440;   // sqrt(2) * 0.5, demo_plasticity_von_mises.py:227
    if constexpr (KIND == DXO_OPERAND_VALUE) {
#pragma unroll
        for (int i = 0; i < BS; ++i) o[i] = val[i];
    } else if constexpr (KIND == DXO_OPERAND_GRAD) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) o[i * G + j] = g[i][j];
    } else if constexpr (KIND == DXO_OPERAND_EPS_MANDEL) {
        if constexpr (G == 2) {
            o[0] = g[0][0]; o[1] = g[1][1]; o[2] = 0.0; o[3] = r2 * (g[0][1] + g[1][0]);
        } else {
            o[0] = g[0][0]; o[1] = g[1][1]; o[2] = g[2][2];
            o[3] = r2 * (g[0][1] + g[1][0]); o[4] = r2 * (g[0][2] + g[2][0]); o[5] = r2 * (g[1][2] + g[2][1]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) o[i * G + j] = g[i][j] + (i == j ? 1.0 : 0.0);
    }
}

// One wave-group of cells: gather -> per-lane gradient -> `o` (D values of this lane's point). Returns false for
// lanes without a point. Shared by the standalone kernel below and by kernels that consume the operand in place.
template <int G, int BS, int KIND>
__device__ __forceinline__ bool operand_point(const OperandDev& m, const double* tab, double* W,
                                              const double* __restrict__ u, const int32_t* __restrict__ cells,
                                              int64_t c0, int ncell, int lane,
                                              double (&o)[OperandShape<G, BS, KIND>::D]) {
    const int nd = m.ndofs, ng = m.ngeom;
    double* U = W;                                   // [ncell][nd][BS]
    double* X = W + m.cells_per_wave * nd * BS;      // [ncell][ng][G]
    // ---- cooperative gather
    for (int idx = lane; idx < ncell * nd; idx += DXO_WAVE) {
        const int c = idx / nd, a = idx - c * nd;
        const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
        const int64_t node = m.dofmap[cell * nd + a];
#pragma unroll
        for (int i = 0; i < BS; ++i) U[idx * BS + i] = u[node * BS + i];
    }
    for (int idx = lane; idx < ncell * ng; idx += DXO_WAVE) {
        const int c = idx / ng, v = idx - c * ng;
        const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
        const int64_t node = m.geom_dofmap[cell * ng + v];
#pragma unroll
        for (int j = 0; j < G; ++j) X[idx * G + j] = m.x[node * G + j];
    }
    op_fence();
    const int c = lane / m.nq, q = lane - c * m.nq;
    const bool active = c < ncell;
    if (active) {
        const double* phi = tab + q * nd;
        const double* dphi = tab + m.nq * nd + (q * nd) * G;
        const double* dpsi = tab + m.nq * nd * (1 + G) + (q * ng) * G;
        double J[G][G], K[G][G];
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < G; ++k) J[j][k] = 0.0;
        for (int v = 0; v < ng; ++v) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] += X[(c * ng + v) * G + j] * dpsi[v * G + k];
        }
        invert<G>(J, K);
        double val[BS], gref[BS][G];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            val[i] = 0.0;
#pragma unroll
            for (int k = 0; k < G; ++k) gref[i][k] = 0.0;
        }
        const double* Uc = U + c * nd * BS;
        for (int a = 0; a < nd; ++a) {
            double ua[BS];
#pragma unroll
            for (int i = 0; i < BS; ++i) ua[i] = Uc[a * BS + i];
            if constexpr (KIND == DXO_OPERAND_VALUE) {
                const double ph = phi[a];
#pragma unroll
                for (int i = 0; i < BS; ++i) val[i] += ua[i] * ph;
            } else {
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const double dk = dphi[a * G + k];
#pragma unroll
                    for (int i = 0; i < BS; ++i) gref[i][k] += ua[i] * dk;
                }
            }
        }
        double g[BS][G];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < G; ++k) s += gref[i][k] * K[k][j];   // d/dx_j = sum_k d/dxi_k * dxi_k/dx_j
                g[i][j] = s;
            }
        shape_operand<G, BS, KIND>(val, g, o);
    }
    op_fence();   // W may be reused by the caller
    return active;
}

}  // namespace

struct dxo_mesh {
    int gdim = 0;
    OperandDev dev{};
    int64_t num_cells = 0, num_field_nodes = 0, num_geom_nodes = 0;
    void* blob = nullptr;       // one device allocation holding tables + dofmaps + coordinates
    double* d_u = nullptr;      // staging for host-resident field vectors
    size_t u_cap = 0;
    int32_t* d_cells = nullptr; // staging for host-resident entity lists
    size_t cells_cap = 0;
    double* d_out = nullptr;    // staging for host-resident outputs
    size_t out_cap = 0;
};
