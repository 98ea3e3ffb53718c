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
    int64_t num_cells_fe;          // cells of the mesh: stride of the element-vector layout fe[a][cell][i] (adjoint kernels)
    const double* phi;             // [nq][ndofs]
    const double* dphi;            // [nq][ndofs][G]
    const double* dpsi;            // [nq][ngeom][G]
    const int32_t* dofmap;         // [num_cells][ndofs]
    const int32_t* geom_dofmap;    // [num_cells][ngeom]
    const double* x;               // [num_geom_nodes][G]
    // a field of ANY block size evaluated one component at a time by the scalar (BS = 1) kernels (dxo_eval_operand): component c of node a
    // sits at u[a * u_stride + c] (the launch passes u + c) and its values go to out[point * out_stride + ...] (the launch passes out + the
    // component's offset). 0 = the kernel's own block size / value size, i.e. a dense field and a dense output.
    int u_stride = 0, out_stride = 0;
};


namespace {


__device__ __forceinline__ void op_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G>
__device__ __forceinline__ double invert(const double (&J)[G][G], double (&K)[G][G]) {   // returns det J
    if constexpr (G == 2) {
        const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double idet = 1.0 / det;
        K[0][0] = J[1][1] * idet; K[0][1] = -J[0][1] * idet;
        K[1][0] = -J[1][0] * idet; K[1][1] = J[0][0] * idet;
        return det;
    } else {
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        const double idet = 1.0 / det;
        K[0][0] = c00 * idet; K[1][0] = c01 * idet; K[2][0] = c02 * idet;
        K[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * idet;
        K[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * idet;
        K[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * idet;
        K[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * idet;
        K[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * idet;
        K[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * idet;
        return det;
    }
}

// Value size of the shaped operand.
template <int G, int BS, int KIND>
struct OperandShape {
    static constexpr int D = KIND == DXO_OPERAND_VALUE ? BS
                           : KIND == DXO_OPERAND_VALUE_GRAD ? BS * (1 + G)
                           : KIND == DXO_OPERAND_GRAD ? BS * G
                           : KIND == DXO_OPERAND_EPS_MANDEL ? (G == 2 ? 4 : 6)
                           : (KIND == DXO_OPERAND_I1 || KIND == DXO_OPERAND_DETF || KIND == DXO_OPERAND_DIV) ? 1
                           : G * G;   // DXO_OPERAND_DEFGRAD, DXO_OPERAND_CAUCHY_GREEN
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
    } else if constexpr (KIND == DXO_OPERAND_VALUE_GRAD) {
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            o[i] = val[i];
#pragma unroll
            for (int j = 0; j < G; ++j) o[BS + i * G + j] = g[i][j];
        }
    } else if constexpr (KIND == DXO_OPERAND_EPS_MANDEL) {
        if constexpr (G == 2) {
            o[0] = g[0][0]; o[1] = g[1][1]; o[2] = 0.0; o[3] = r2 * (g[0][1] + g[1][0]);
        } else {
            o[0] = g[0][0]; o[1] = g[1][1]; o[2] = g[2][2];
            o[3] = r2 * (g[0][1] + g[1][0]); o[4] = r2 * (g[0][2] + g[2][0]); o[5] = r2 * (g[1][2] + g[2][1]);
        }
    } else if constexpr (KIND == DXO_OPERAND_DEFGRAD) {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) o[i * G + j] = g[i][j] + (i == j ? 1.0 : 0.0);
    } else if constexpr (KIND == DXO_OPERAND_DIV) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) t += g[i % BS][i];      // BS == G
        o[0] = t;
    } else {
        // nonlinear operands of F = I + grad u (test/test_operands_evaluation.py:32-36); BS == G
        double F[G][G];
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) F[i][j] = g[i % BS][j] + (i == j ? 1.0 : 0.0);
        if constexpr (KIND == DXO_OPERAND_CAUCHY_GREEN) {
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    double c = 0.0;
#pragma unroll
                    for (int k = 0; k < G; ++k) c += F[k][i] * F[k][j];      // (F^T F)_ij
                    o[i * G + j] = c;
                }
        } else if constexpr (KIND == DXO_OPERAND_I1) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int j = 0; j < G; ++j) t += F[i][j] * F[i][j];
            o[0] = t;
        } else {
            if constexpr (G == 2) o[0] = F[0][0] * F[1][1] - F[0][1] * F[1][0];
            else o[0] = F[0][0] * (F[1][1] * F[2][2] - F[1][2] * F[2][1]) - F[0][1] * (F[1][0] * F[2][2] - F[1][2] * F[2][0]) +
                        F[0][2] * (F[1][0] * F[2][1] - F[1][1] * F[2][0]);
        }
    }
}

// ---- LDS layout. Lanes of one wave read, in the same instruction, the SAME slot of up to 64/nq different cells
// (U, X) or of nq different points (tables): a per-cell / per-point stride that is an ODD number of doubles sends
// those addresses to different banks (an even stride such as 24 doubles = 48 dwords collides 4-way on 32 banks).
__device__ __host__ __forceinline__ int op_odd(int n) { return n | 1; }

template <int G>
struct OperandLayout {
    int sphi, sdphi, sdpsi;     // per-point strides of the three tables
    int o_dphi, o_dpsi;         // table offsets
    int sx;                     // per-cell stride of the coordinates
    __device__ __host__ OperandLayout(const OperandDev& m)
        : sphi(op_odd(m.ndofs)), sdphi(op_odd(m.ndofs * G)), sdpsi(op_odd(m.ngeom * G)),
          o_dphi(m.nq * op_odd(m.ndofs)), o_dpsi(m.nq * (op_odd(m.ndofs) + op_odd(m.ndofs * G))), sx(op_odd(m.ngeom * G)) {}
};

// tables -> LDS in the padded layout (whole workgroup, followed by __syncthreads() in the caller)
template <int G>
__device__ __forceinline__ void operand_load_tables(const OperandDev& m, double* tab) {
    const OperandLayout<G> L(m);
    for (int i = threadIdx.x; i < m.nq * m.ndofs; i += blockDim.x) {
        const int q = i / m.ndofs;
        tab[q * L.sphi + (i - q * m.ndofs)] = m.phi[i];
    }
    for (int i = threadIdx.x; i < m.nq * m.ndofs * G; i += blockDim.x) {
        const int q = i / (m.ndofs * G);
        tab[L.o_dphi + q * L.sdphi + (i - q * m.ndofs * G)] = m.dphi[i];
    }
    for (int i = threadIdx.x; i < m.nq * m.ngeom * G; i += blockDim.x) {
        const int q = i / (m.ngeom * G);
        tab[L.o_dpsi + q * L.sdpsi + (i - q * m.ngeom * G)] = m.dpsi[i];
    }
}

// ---- which wave-groups a wave walks. Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has
// its own L2, while a mesh node is gathered by every cell that touches it (3.4 times on average for Q2 hexahedra). So
// each XCD gets ONE contiguous eighth of the groups (a slab of the mesh) and its workgroups sweep that slab together:
// the re-reads of shared nodes then hit the XCD's L2 instead of crossing to another XCD's or to HBM.
struct GroupWalk {
    int64_t first, end, stride;
};
__device__ __forceinline__ GroupWalk xcd_group_walk(int64_t n_groups, int waves_per_block, int wave) {
    constexpr int XCDS = 8;
    if (gridDim.x % XCDS != 0) return {(int64_t)blockIdx.x * waves_per_block + wave, n_groups, (int64_t)gridDim.x * waves_per_block};
    const int xcd = blockIdx.x % XCDS, local = blockIdx.x / XCDS, per_xcd = gridDim.x / XCDS;
    const int64_t begin = n_groups * xcd / XCDS, end = n_groups * (xcd + 1) / XCDS;
    return {begin + (int64_t)local * waves_per_block + wave, end, (int64_t)per_xcd * waves_per_block};
}

// ---- gather of one wave-group's dofs and vertex coordinates into the wave's LDS buffer W
template <int G, int BS>
__device__ __forceinline__ void operand_gather(const OperandDev& m, double* W, const double* __restrict__ u,
                                               const int32_t* __restrict__ cells, int64_t c0, int ncell, int lane) {
    const int nd = m.ndofs, ng = m.ngeom;
    const int su = op_odd(nd * BS), sx = op_odd(ng * G);
    double* U = W;                                   // [ncell] x su : [nd][BS]
    double* X = W + m.cells_per_wave * su;           // [ncell] x sx : [ng][G]
    for (int idx = lane; idx < ncell * nd; idx += DXO_WAVE) {
        const int c = idx / nd, a = idx - c * nd;
        const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
        const int64_t node = m.dofmap[cell * nd + a];
#pragma unroll
        for (int i = 0; i < BS; ++i) U[c * su + a * BS + i] = u[node * (m.u_stride ? m.u_stride : BS) + i];
    }
    for (int idx = lane; idx < ncell * ng; idx += DXO_WAVE) {
        const int c = idx / ng, v = idx - c * ng;
        const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
        const int64_t node = m.geom_dofmap[cell * ng + v];
#pragma unroll
        for (int j = 0; j < G; ++j) X[c * sx + v * G + j] = m.x[node * G + j];
    }
}

// ---- the same gather as a two-deep register pipeline (entity list absent, at most OP_GI*64 dof slots and
// OP_XI*64 vertex slots per wave-group). While group g is being computed from LDS, the values of group g+1 are in
// flight into `ud/xd` and the node indices of group g+2 into `un/xn`, so neither the dofmap -> u dependency nor the
// HBM/L2 latency of the scattered 8-byte loads is exposed inside a wave.
constexpr int OP_GI = 4, OP_XI = 2;

template <int G, int BS>
struct OperandPipe {
    int32_t un[OP_GI], xn[OP_XI];
    double ud[OP_GI][BS], xd[OP_XI][G];
};

__device__ __forceinline__ bool operand_can_pipe(const OperandDev& m) {
    return m.cells_per_wave * m.ndofs <= OP_GI * DXO_WAVE && m.cells_per_wave * m.ngeom <= OP_XI * DXO_WAVE;
}

template <int G, int BS>
__device__ __forceinline__ void pipe_load_indices(const OperandDev& m, OperandPipe<G, BS>& pf, int64_t c0, int ncell, int lane) {
    const int nd = m.ndofs, ng = m.ngeom;
#pragma unroll
    for (int it = 0; it < OP_GI; ++it) {
        const int idx = it * DXO_WAVE + lane;
        pf.un[it] = -1;
        if (idx < ncell * nd) pf.un[it] = m.dofmap[c0 * nd + idx];          // cells are consecutive: one flat slice
    }
#pragma unroll
    for (int it = 0; it < OP_XI; ++it) {
        const int idx = it * DXO_WAVE + lane;
        pf.xn[it] = -1;
        if (idx < ncell * ng) pf.xn[it] = m.geom_dofmap[c0 * ng + idx];
    }
}

template <int G, int BS>
__device__ __forceinline__ void pipe_load_values(const OperandDev& m, OperandPipe<G, BS>& pf, const double* __restrict__ u) {
#pragma unroll
    for (int it = 0; it < OP_GI; ++it)
        if (pf.un[it] >= 0) {
#pragma unroll
            for (int i = 0; i < BS; ++i) pf.ud[it][i] = u[(int64_t)pf.un[it] * (m.u_stride ? m.u_stride : BS) + i];
        }
#pragma unroll
    for (int it = 0; it < OP_XI; ++it)
        if (pf.xn[it] >= 0) {
#pragma unroll
            for (int j = 0; j < G; ++j) pf.xd[it][j] = m.x[(int64_t)pf.xn[it] * G + j];
        }
}

template <int G, int BS>
__device__ __forceinline__ void pipe_commit(const OperandDev& m, const OperandPipe<G, BS>& pf, double* W, int ncell, int lane) {
    const int nd = m.ndofs, ng = m.ngeom;
    const int su = op_odd(nd * BS), sx = op_odd(ng * G);
    double* U = W;
    double* X = W + m.cells_per_wave * su;
#pragma unroll
    for (int it = 0; it < OP_GI; ++it) {
        const int idx = it * DXO_WAVE + lane;
        if (idx < ncell * nd) {
            const int c = idx / nd, a = idx - c * nd;
#pragma unroll
            for (int i = 0; i < BS; ++i) U[c * su + a * BS + i] = pf.ud[it][i];
        }
    }
#pragma unroll
    for (int it = 0; it < OP_XI; ++it) {
        const int idx = it * DXO_WAVE + lane;
        if (idx < ncell * ng) {
            const int c = idx / ng, v = idx - c * ng;
#pragma unroll
            for (int j = 0; j < G; ++j) X[c * sx + v * G + j] = pf.xd[it][j];
        }
    }
}

// ---- per-lane gradient from the gathered data -> `o` (D values of this lane's point). Returns false for lanes
// without a point. Shared by the standalone kernel and by kernels that consume the operand in place.
#ifndef DXO_OP_UNROLL
#define DXO_OP_UNROLL 3
#endif
// ND_CT / NG_CT: nodes / vertices per cell known at compile time (0 = read them from the mesh descriptor): constant
// trip counts let the compiler address the LDS tables with immediates and pipeline the reads across the whole loop.
template <int G, int BS, int KIND, int ND_CT = 0, int NG_CT = 0>
__device__ __forceinline__ bool operand_compute_geo(const OperandDev& m, const double* tab, double* W, int ncell, int lane,
                                                    double (&o)[OperandShape<G, BS, KIND>::D], double (&K)[G][G],
                                                    double& detJ) {
    const int nd = ND_CT ? ND_CT : m.ndofs, ng = NG_CT ? NG_CT : m.ngeom;
    const OperandLayout<G> L(m);
    const int su = op_odd(nd * BS);
    double* U = W;
    double* X = W + m.cells_per_wave * su;
    op_fence();
    const int c = lane / m.nq, q = lane - c * m.nq;
    const bool active = c < ncell;
    if (active) {
        const double* phi = tab + q * L.sphi;
        const double* dphi = tab + L.o_dphi + q * L.sdphi;
        const double* dpsi = tab + L.o_dpsi + q * L.sdpsi;
        const double* Xc = X + c * L.sx;
        double J[G][G];
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < G; ++k) J[j][k] = 0.0;
#pragma unroll 2
        for (int v = 0; v < ng; ++v) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] += Xc[v * G + j] * dpsi[v * G + k];
        }
        detJ = invert<G>(J, K);
        double val[BS], gref[BS][G];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            val[i] = 0.0;
#pragma unroll
            for (int k = 0; k < G; ++k) gref[i][k] = 0.0;
        }
        const double* Uc = U + c * su;
        // unrolled (fully when the node count is a compile-time constant, else by three): batches the LDS reads of several
        // nodes so their latency overlaps the FMAs of the previous ones. Measured on vm_field<3>, Q2 hexahedra, 10^7 points
        // (scripts/exp/vmfield_ab.py): run-time count x3 1.026 ms, compile-time x3 1.002, x9 1.001, x27 0.973.
#pragma unroll(ND_CT ? ND_CT : DXO_OP_UNROLL)
        for (int a = 0; a < nd; ++a) {
            double ua[BS];
#pragma unroll
            for (int i = 0; i < BS; ++i) ua[i] = Uc[a * BS + i];
            if constexpr (KIND == DXO_OPERAND_VALUE || KIND == DXO_OPERAND_VALUE_GRAD) {
                const double ph = phi[a];
#pragma unroll
                for (int i = 0; i < BS; ++i) val[i] += ua[i] * ph;
            }
            if constexpr (KIND != DXO_OPERAND_VALUE) {
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

template <int G, int BS, int KIND, int ND_CT = 0, int NG_CT = 0>
__device__ __forceinline__ bool operand_compute(const OperandDev& m, const double* tab, double* W, int ncell, int lane,
                                                double (&o)[OperandShape<G, BS, KIND>::D]) {
    double K[G][G], detJ;
    return operand_compute_geo<G, BS, KIND, ND_CT, NG_CT>(m, tab, W, ncell, lane, o, K, detJ);
}

// gather + compute for one group (no pipelining): entity lists and elements too large for the register pipeline
template <int G, int BS, int KIND>
__device__ __forceinline__ bool operand_point(const OperandDev& m, const double* tab, double* W,
                                              const double* __restrict__ u, const int32_t* __restrict__ cells,
                                              int64_t c0, int ncell, int lane,
                                              double (&o)[OperandShape<G, BS, KIND>::D]) {
    operand_gather<G, BS>(m, W, u, cells, c0, ncell, lane);
    return operand_compute<G, BS, KIND>(m, tab, W, ncell, lane, o);
}

}  // namespace

#ifndef DXO_PATCH_WAVES
#define DXO_PATCH_WAVES 4     // waves of a patch workgroup = wave groups per iteration
#endif
constexpr int PATCH_WAVES = DXO_PATCH_WAVES;
constexpr int PATCH_BLOCK = PATCH_WAVES * DXO_WAVE;

// what the kernels read (device pointers, owned by dxo_mesh::patch)
struct PatchDev {
    int32_t n_patches = 0;
    int32_t R = 0;                       // wave groups per wave and patch
    int32_t max_priv = 0;                // largest number of nodes in one wave's accumulator
    int32_t max_local = 0;               // largest number of nodes of a patch
    const int32_t* groups = nullptr;     // [n_patches][PATCH_WAVES][R] wave group, -1: none
    const int32_t* node_off = nullptr;   // [n_patches + 1] offsets of the patches' nodes (= slots of bpart)
    const uint32_t* gnode = nullptr;     // [n_slots] global node of a patch node; bit 31: shared with another patch
    const uint16_t* mmap = nullptr;      // [n_slots][PATCH_WAVES] the node's entry in wave w's accumulator, 0xffff: none
    const uint16_t* lnode = nullptr;     // [num_cells][ndofs] entry of (cell, local dof) in the accumulator of the wave that owns the cell
    const uint8_t* cellcol = nullptr;    // [num_cells] colour of the cell within its wave group
    const uint8_t* grp_ncol = nullptr;   // [n_groups] number of cell colours of the group
    double* bpart = nullptr;             // [n_slots][bs] partials of the shared nodes (slots of unshared nodes unused)
    // second pass
    int64_t n_bnodes = 0;
    const int32_t* bnode = nullptr;      // [n_bnodes] shared (or untouched) nodes
    const int64_t* bptr = nullptr;       // [n_bnodes + 1]
    const uint32_t* bent = nullptr;      // slots of the node's partials, ascending patch order (+ 3 entries of padding)
};

struct PatchSet {
    bool built = false, usable = false;
    int cpw = 0, bs_cap = 0;
    PatchDev dev;
    void* blob = nullptr;                // one device allocation for the index arrays
    size_t bpart_cap = 0;
    int64_t n_slots = 0;
    double shared_fraction = 0.0;        // shared patch nodes / (cell, node) incidences: what still travels through HBM
};

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
    double* d_psi = nullptr;    // [nq][ngeom] values of the coordinate element at the quadrature points (dxo_mesh_set_coordinate_values)
    double* d_wq = nullptr;     // quadrature weights (dxo_mesh_set_weights), needed by the adjoint kernels only
    // codim-1 evaluation (dxo_mesh_set_facet_tables / dxo_eval_operand_facets): tables per LOCAL facet of the cell
    int n_local_facets = 0, nq_facet = 0;
    double* d_facet_tab = nullptr;      // phi_f [nf][nqf][ndofs] | dphi_f [nf][nqf][ndofs][G] | dpsi_f [nf][nqf][ngeom][G]
    int32_t* d_ents = nullptr;          // staging for host-resident (cell, local facet) lists
    size_t ents_cap = 0;
    // adjoint kernels, two-pass form: element vectors + the transposed dofmap (node -> its (cell, local node) entries)
    std::vector<int32_t> h_dofmap;     // host copy kept for building the transpose on first use
    int64_t* d_node_ptr = nullptr;     // [num_field_nodes + 1]
    uint32_t* d_node_ent = nullptr;    // [num_cells * ndofs], values a * num_cells + cell (index into d_fe), fixed order per node
    double* d_fe = nullptr;            // [ndofs][num_cells][bs] element vectors of the last adjoint call
    size_t fe_cap = 0;
    // adjoint kernels, patch form (adjoint_patch.h): a representative point per cell for the Morton order of the wave groups
    std::vector<float> h_cell_xyz;     // [num_cells][3]
    PatchSet patch;
};
