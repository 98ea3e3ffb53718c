// cell8_dpp.h — operand contraction and its transpose for cells of EIGHT quadrature points and at most 32 nodes (Q2 / Q1
// hexahedra with the 2x2x2 rule), carried out across the 8 lanes of a cell with DPP lane permutations instead of LDS.
//
// Reference arithmetic: the push-forward of fem.Expression(eps(u), points).eval (src/dolfinx_external_operator/
// external_operator.py:386-402) and its adjoint, the assembly of inner(s, eps(v)) dx (:463-486) — the same sums as
// operand_core.h / adjoint.hip form through LDS, in another order.
//
// Layout: lane = (cell c = lane >> 3, point q = lane & 7); a wave holds 8 cells. Lane q OWNS the four nodes
// M(q) .. M(q)+3 of its cell, M(q) = 28 b2 ^ 8 b1 ^ 4 b0 (b2 b1 b0 the bits of q): it gathers their dof values straight from
// global memory into registers (no staging in LDS) and it is the lane that ends up with their element-vector entries.
//   forward   grad_ref u (q) = sum_a u_a (x) dphi_a(q): for every group t = 0..3 of nodes the 8 lanes ALL-GATHER one value each
//             (7 DPP exchanges: q^1, q^2 by quad_perm, q^7 by row_half_mirror) and every lane multiplies the 8 values it now
//             holds with ITS point's table rows. Register j = (h, m, l) of lane q holds node (16h + 8m + 4l + t) ^ M(q) — a
//             lane-dependent table row, but a fixed exchange pattern without selects, because M(q^1) ^ M(q) = 4,
//             M(q^2) ^ M(q) = 8 and M(q^7) ^ M(q) = 16.
//   transpose f_a = sum_q T_q dphi_a(q): every lane forms its point's partial for the same 8 nodes and the 8 lanes
//             REDUCE-SCATTER them (the exchanges of the forward pass, backwards, with an add): 7 exchanges per 8 partials instead
//             of the 24 of a butterfly; the order of the additions is fixed (bit-reproducible).
// The table is padded to 32 nodes x 4 doubles per point with zeros, so nodes >= ndofs contribute nothing and are not stored.
// LDS traffic per lane and wave group: 64 table reads per pass (plus 24 for the geometry) instead of 210 (forward) and 393
// (transpose) in the LDS forms — the consumer kernels were bound by LDS wave-instructions (profiles/r04_adjoint_experiments.txt).
#pragma once

#include "dxo_common.h"
#include "operand_core.h"

namespace {

constexpr int C8_NODES = 32;
// per-point stride of the dphi table: 32 rows of 4 doubles + 2. The 8 lanes of a cell read 8 different rows in one instruction —
// row (4j + t) ^ M(q) of point q — and with the natural stride (256 dwords = 0 mod 64 banks) those rows fall on two bank
// positions only (M(q) mod 8 is 0 or 4): a 4-way conflict, a third to a half of the kernels' LDS cycles
// (profiles/r04_device_loop_pmc.json). With 4 dwords more per point the 8 reads land on 8 different 4-bank slots.
constexpr int C8_QSTRIDE = C8_NODES * 4 + 2;
constexpr int C8_TAB = 8 * C8_QSTRIDE;          // doubles: [q][node][dphi_x, dphi_y, dphi_z, 0] (+ 2 per q)
constexpr int C8_GEO = 8 * 8 * 4;               // doubles: [q][vertex][dpsi_x, dpsi_y, dpsi_z, 0]
constexpr int C8_LDS = C8_TAB + C8_GEO;
constexpr int C8_HALF_MIRROR = 0x141, C8_XOR2 = 0x4E, C8_XOR1 = 0xB1;   // row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2]

template <int CTRL>
__device__ __forceinline__ double c8_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// one value per lane -> the 8 values of the cell's lanes in register order j = (h, m, l): u[j] is the value of lane q ^ X(j)
// with X(1) = 1, X(2) = 2, X(3) = 3, X(4 + jj) = 7 ^ X(jj)
__device__ __forceinline__ void c8_all_gather(double x, double (&u)[8]) {
    u[0] = x;
    u[1] = c8_dpp<C8_XOR1>(u[0]);
    u[2] = c8_dpp<C8_XOR2>(u[0]);
    u[3] = c8_dpp<C8_XOR2>(u[1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) u[4 + j] = c8_dpp<C8_HALF_MIRROR>(u[j]);
}

// the transpose: partials p[j] of the 8 register slots -> the sum over the cell's 8 lanes of the slot THIS lane owns
__device__ __forceinline__ double c8_reduce_scatter(double (&p)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] += c8_dpp<C8_HALF_MIRROR>(p[4 + j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) p[j] += c8_dpp<C8_XOR2>(p[2 + j]);
    return p[0] + c8_dpp<C8_XOR1>(p[1]);
}

// whole workgroup, followed by __syncthreads() in the caller
__device__ __forceinline__ void c8_fill_tables(const OperandDev& m, double* tabP) {
    for (int e = threadIdx.x; e < 8 * C8_NODES * 4; e += blockDim.x) {
        const int k = e & 3, a = (e >> 2) & (C8_NODES - 1), q = e >> 7;
        tabP[q * C8_QSTRIDE + a * 4 + k] = (k < 3 && a < m.ndofs) ? m.dphi[(q * m.ndofs + a) * 3 + k] : 0.0;
    }
    for (int e = threadIdx.x; e < C8_GEO; e += blockDim.x) {
        const int k = e & 3, v = (e >> 2) & 7, q = e >> 5;
        tabP[C8_TAB + e] = k < 3 ? m.dpsi[(q * 8 + v) * 3 + k] : 0.0;
    }
}

struct C8Lane {
    const double* tabq;       // this point's dphi rows: tabq + 4 node
    const double* geoq;       // this point's dpsi rows: geoq + 4 vertex
    int node0, q;             // M(q), q
    __device__ __forceinline__ C8Lane(const double* tabP, int lane) {
        q = lane & 7;
        node0 = ((q & 4) ? 28 : 0) ^ ((q & 2) ? 8 : 0) ^ ((q & 1) ? 4 : 0);
        tabq = tabP + q * C8_QSTRIDE;
        geoq = tabP + C8_TAB + q * 8 * 4;
    }
    // dphi row of register j in group t: node (4 j + t) ^ M(q)  (addresses are formed where they are used: two integers per lane
    // instead of sixteen pointers)
    __device__ __forceinline__ const double* row(int j, int t) const { return tabq + ((((4 * j) ^ node0) + t) << 2); }
    // dpsi row of the vertex held by register j after an all-gather: vertex q ^ X(j)
    __device__ __forceinline__ const double* geo(int j) const { return geoq + ((q ^ ((j & 4) ? (7 ^ (j & 3)) : j)) << 2); }
};

// ---- gather pipeline in registers: the dof values of the lane's four nodes and the coordinates of vertex q of its cell
struct C8Pipe {
    int32_t un[4], xn;
    double ud[4][3], xd[3];
};

template <int ND>
__device__ __forceinline__ void c8_load_indices(const OperandDev& m, C8Pipe& pf, const C8Lane& L, int64_t c0, int ncell, int lane) {
    const int c = lane >> 3, q = lane & 7;
    const bool has = c < ncell;
    const int64_t cell = c0 + c;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int a = L.node0 + t;
        pf.un[t] = (has && a < ND) ? m.dofmap[cell * ND + a] : -1;
    }
    pf.xn = has ? m.geom_dofmap[cell * 8 + q] : -1;
}

__device__ __forceinline__ void c8_load_values(const OperandDev& m, C8Pipe& pf, const double* __restrict__ u) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i) pf.ud[t][i] = pf.un[t] >= 0 ? u[(int64_t)pf.un[t] * 3 + i] : 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) pf.xd[j] = pf.xn >= 0 ? m.x[(int64_t)pf.xn * 3 + j] : 0.0;
}

// J^-1 and det J of this lane's point from the cell's 8 vertices (one per lane)
__device__ __forceinline__ double c8_geometry(const C8Lane& L, const double (&xv)[3], double (&K)[3][3]) {
    double J[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) J[j][k] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double xs[8];
        c8_all_gather(xv[j], xs);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const dxo_f64x2 a = *reinterpret_cast<const dxo_f64x2*>(L.geo(v));
            const double b = L.geo(v)[2];
            J[j][0] += xs[v] * a.x; J[j][1] += xs[v] * a.y; J[j][2] += xs[v] * b;
        }
    }
    return invert<3>(J, K);
}

// reference gradient gref[i][k] = sum_a U_a,i dphi_a,k(q) of this lane's point; U4[t][i]: the dof values of the lane's own nodes
__device__ __forceinline__ void c8_forward(const C8Lane& L, const double (&U4)[4][3], double (&gref)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) gref[i][k] = 0.0;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        double d[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* r = L.row(j, t);
            const dxo_f64x2 a = *reinterpret_cast<const dxo_f64x2*>(r);
            d[j][0] = a.x; d[j][1] = a.y; d[j][2] = r[2];
        }
        // U4[t] with a run-time t: select instead of indexing (registers cannot be indexed)
        double own[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) own[i] = t == 0 ? U4[0][i] : t == 1 ? U4[1][i] : t == 2 ? U4[2][i] : U4[3][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double u[8];
            c8_all_gather(own[i], u);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) gref[i][k] += u[j] * d[j][k];
        }
    }
}

// transpose: T[i][k] of this lane's point (zero for lanes without one) -> element vector of the cell, entries of the lane's
// own nodes, written to fe[node][cell][i] (two-pass form) or added to out through the dofmap (atomics)
// UT: how many of the four node groups are in flight together (each holds 8 table rows + 8 partials: 64 registers)
template <int ND, int UT = 1, typename Store>
__device__ __forceinline__ void c8_scatter(const C8Lane& L, const double (&T)[3][3], Store&& store) {
#pragma unroll UT
    for (int t = 0; t < 4; ++t) {
        double d[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* r = L.row(j, t);
            const dxo_f64x2 a = *reinterpret_cast<const dxo_f64x2*>(r);
            d[j][0] = a.x; d[j][1] = a.y; d[j][2] = r[2];
        }
        double o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = T[i][0] * d[j][0] + T[i][1] * d[j][1] + T[i][2] * d[j][2];
            o[i] = c8_reduce_scatter(p);
        }
        const int a = L.node0 + t;
        if (a < ND) store(a, o);
    }
}

// the same with a store every lane calls for every node group t (the patch form adds into LDS in passes that need uniform
// control flow; the store masks the lanes without a node itself)
template <int ND, typename Store>
__device__ __forceinline__ void c8_scatter_all(const C8Lane& L, const double (&T)[3][3], Store&& store) {
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        double d[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double* r = L.row(j, t);
            const dxo_f64x2 a = *reinterpret_cast<const dxo_f64x2*>(r);
            d[j][0] = a.x; d[j][1] = a.y; d[j][2] = r[2];
        }
        double o[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = T[i][0] * d[j][0] + T[i][1] * d[j][1] + T[i][2] * d[j][2];
            o[i] = c8_reduce_scatter(p);
        }
        store(t, o);
    }
}

}  // namespace
