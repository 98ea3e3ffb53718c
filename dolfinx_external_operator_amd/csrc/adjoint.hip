// adjoint.hip — the consumer side of the path on the device: virtual work of a quadrature field and
// the matrix-free action of the tangent.
//
// In the reference the coefficient the operator wrote is consumed by DOLFINx assembly: the residual form
// inner(sigma, eps(v)) dx and the Jacobian form inner(C_tang : eps(du), eps(v)) dx that `_apply_derivative_tensor`
// produces (src/dolfinx_external_operator/external_operator.py:463-486; demo_plasticity_von_mises.py:378-391). That
// step is what forces the 16/36-double tangents across PCIe. Its arithmetic is the ADJOINT of the operand evaluation:
//   operand      e_q   = B_q u          (gather dofs, contract with grad phi, push forward, shape)      operand.hip
//   virtual work f     = sum_q w_q |det J_q| B_q^T s_q                                                  dxo_operand_adjoint
//   tangent      K v   = sum_q w_q |det J_q| B_q^T C_q B_q v   (never forming K)                        dxo_tangent_apply[_vm]
//   its diagonal                                                                                        dxo_tangent_diagonal[_vm]
// so with these entry points a Newton-Krylov solver can keep sigma and C_tang in HBM and move only dof vectors.
//
// Shape of every call (whole mesh): TWO PASSES, no atomics, bit-reproducible.
//   pass 1, element kernel: a wave owns floor(64 / nq) consecutive cells, lane = (cell, point). J^-1 and det J from the
//     gathered vertices; the point's dual tensor pulled back to reference gradients T_q[i][k] = w |det J| sum_j G_ij K[k][j];
//     the element-vector entries f_a,i = sum_q sum_k T_q[i][k] dphi_a,k(xi_q) are written to fe[local node][cell][i].
//     Q2 / Q1 hexahedra with the 2x2x2 rule form the entries of a wave's 8 cells on the fp64 matrix pipe, f[a][(c, i)] =
//     D[a][(q, k)] T[(q, k)][(c, i)] as 24 v_mfma_f64_16x16x4_f64 (c8m_contract: operand_adjoint_c8_mfma, tangent_apply<3, 27, 8, .., MF>;
//     option adjoint_mfma = 0: the round-4 form, everything in registers and a DPP reduce-scatter over a cell's 8 lanes, cell8_dpp.h:
//     operand_adjoint_c8, tangent_apply<3, 27, 8>; round 5, Q2 hexahedra 108^3: state-based action 0.88 -> 0.80 ms, C_tang rows 1.13 ->
//     1.10, internal force 0.61 -> 0.59); the other elements park T_q in the
//     wave's LDS slice and let lane = (cell, node) form the entries (adjoint_scatter); P2 triangles' internal force has a
//     lane = cell form (adjoint_cell.h). tangent_apply requests its tangent rows lane-linear and passes them through LDS
//     (TangentRows); the *_vm forms rebuild the tangent's action from the returned (sigma, dp) instead (VmStateSrc).
//   pass 2, node_sum: one thread per node adds the node's entries in the fixed order of the transposed dofmap (built once per
//     mesh) — or SETS the vector to the sums with option consumer_overwrite (no memset before a Krylov matvec).
// Entity subsets, or option adjoint_atomics = 1, add with fp64 hardware atomics into the dof vector instead (one pass,
// reproducible to rounding only). The hexahedral internal force with the entries added in LDS patch by patch (option adjoint_patch,
// measured slower; profiles/r05_patch_form.txt) lives in scripts/exp/adjoint_patch*.h and is compiled only with -DDXO_EXPERIMENTS.
// Kernel variants that were measured and not shipped (contraction across the lanes as well, lane = cell tangent action) live in
// scripts/exp/adjoint_variants.h and are compiled only with -DDXO_EXPERIMENTS.
#include "dxo_common.h"
#include "operand_core.h"
#include "adjoint_cell.h"
#include "cell8_dpp.h"
#include "cell8_mfma.h"
#include "scatter_mfma.h"
#include "vm_core.h"
#ifdef DXO_EXPERIMENTS
#include "../../scripts/exp/adjoint_patch.h"
#endif

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <atomic>

// experiment switches (scripts/exp/ab_adjoint.py builds variants with -D...)
#ifndef DXO_TA_NT
#define DXO_TA_NT 0          // non-temporal loads of the tangent rows
#endif
#ifndef DXO_TA_PIPE
#define DXO_TA_PIPE 1        // register-pipelined dof gather
#endif
#ifndef DXO_TA_EARLY_C
#define DXO_TA_EARLY_C 1     // request the tangent row before the contraction
#endif
#ifndef DXO_ADJ_PAD
#define DXO_ADJ_PAD 1        // odd per-point stride of the parked tensors
#endif
#ifndef DXO_TA_WAVES
#define DXO_TA_WAVES 2
#endif
#ifndef DXO_TA_RS
#define DXO_TA_RS 1          // Q2 hexahedra: scatter phase in registers with a DPP reduce-scatter (scatter_rs)
#endif
#ifdef DXO_EXPERIMENTS       // variants that were measured and not shipped (scripts/exp/adjoint_variants.h)
#ifndef DXO_TA_C8_FORWARD
#define DXO_TA_C8_FORWARD 0  // tangent_apply_c8 (contraction across the lanes as well)
#endif
#ifndef DXO_TANGENT_CELL
#define DXO_TANGENT_CELL 0   // lane = cell tangent action on P2 triangles
#endif
#ifndef DXO_C8_EARLY_C
#define DXO_C8_EARLY_C 0     // tangent_apply_c8: 0 = the tangent rows are requested after the contraction (72 registers the pass does not have)
#endif
#else
#define DXO_TA_C8_FORWARD 0
#define DXO_TANGENT_CELL 0
#endif
#ifndef DXO_TA_GM_WAVES
#define DXO_TA_GM_WAVES 2    // P2 tetrahedra / triangles with the MFMA scatter (scatter_mfma.h). Tetrahedra: 196 registers; at three waves per SIMD 13 are
                             // spilled and the matvec takes 0.683 instead of 0.661 ms (LDS form 0.74-0.76); triangles fit three waves either way (148)
#endif
#ifndef DXO_TA_VM_MF_WAVES
#define DXO_TA_VM_MF_WAVES 2 // the same with the MFMA scatter (option adjoint_mfma)
#endif
#ifndef DXO_TA_VM_WAVES
#define DXO_TA_VM_WAVES 2    // waves per SIMD of the state-based tangent action (3: 45 registers spilled on hexahedra)
#endif
#ifndef DXO_TA_STAGE
#define DXO_TA_STAGE 1       // tangent rows requested lane-linear and passed through LDS (TangentRows) instead of row-per-lane loads
#endif

namespace {

template <bool NT>
__device__ __forceinline__ dxo_f64x2 ta_load(const dxo_f64x2* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// ---- the tangent rows of a wave group, coalesced. A lane needs the D*D entries of ITS point; asking for them row-per-lane makes
// every load instruction touch 64 different cache lines, 16 bytes of each, and with ~150 KB of such rows in flight per CU the
// lines are evicted from the 32 KB L1 between the instructions that use their other pieces (tangent_apply ran at a third of the
// rate its traffic allows). So the group's rows are requested LANE-LINEAR (16 bytes per lane, consecutive lanes consecutive
// addresses: every line is fetched once) at the top of the iteration and kept in registers; once the wave's LDS region is free
// they pass through it in two chunks of TR_PC points and every lane picks up its own row.
constexpr int TR_PC = 32;                       // points per chunk
template <int D>
struct TangentRows {
    static constexpr int CV = D * D / 2;        // 16-byte units per point
    static constexpr int LPC = TR_PC * CV / DXO_WAVE;   // lane-linear loads per lane per chunk (9 at d = 6, 4 at d = 4)
    // a staged row occupies CV + 1 units: lanes read their rows 16 bytes at a time at a stride of one row, and with the natural
    // strides (128 bytes at d = 4: 32 dwords; 288 at d = 6: 8 mod 64 dwords) the 16 lanes of a read group hit two / eight bank
    // positions — an 8-way and a 2-way conflict (28-32 % of the triangle kernels' LDS cycles, profiles/r04_device_loop_pmc.json);
    // 144 / 304 bytes put the 16 rows on 16 different 4-bank slots
    static constexpr int RS = CV + 1;
    static constexpr int LDS_DOUBLES = TR_PC * RS * 2;  // staging space one chunk needs
    static __device__ __forceinline__ int slot(int u) { return (u / CV) * RS + (u % CV); }   // u-th 16-byte unit of the chunk
    dxo_f64x2 r[2][LPC];
    __device__ __forceinline__ void request(const double* __restrict__ C_tang, int64_t p0, int npts, int lane) {
        const dxo_f64x2* base = reinterpret_cast<const dxo_f64x2*>(C_tang + p0 * (D * D));
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < LPC; ++k) {
                const int u = c * TR_PC * CV + k * DXO_WAVE + lane;
                r[c][k] = u < npts * CV ? ta_load<DXO_TA_NT != 0>(base + u) : dxo_f64x2{1.0, 0.5};
            }
    }
    // t = C_row e for this lane's point straight from the staged rows (the row never sits in registers)
    __device__ __forceinline__ void times(double* S, int lane, const double (&e)[D], double (&t)[D]) const {
        dxo_f64x2* S2 = reinterpret_cast<dxo_f64x2*>(S);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int k = 0; k < LPC; ++k) S2[slot(k * DXO_WAVE + lane)] = r[c][k];
            op_fence();
            if (lane / TR_PC == c) {
                const dxo_f64x2* R = S2 + (lane - c * TR_PC) * RS;
#pragma unroll
                for (int rr = 0; rr < D; ++rr) {
                    double acc = 0.0;
#pragma unroll
                    for (int cc = 0; cc < D; cc += 2) {
                        const dxo_f64x2 c2 = R[(rr * D + cc) / 2];
                        acc += c2.x * e[cc];
                        acc += c2.y * e[cc + 1];
                    }
                    t[rr] = acc;
                }
            }
            op_fence();
        }
    }
    // -> row[CV] of this lane's point (lane = point index inside the group); S = the wave's LDS region (16-byte aligned)
    __device__ __forceinline__ void deliver(double* S, int lane, dxo_f64x2 (&row)[CV]) const {
        dxo_f64x2* S2 = reinterpret_cast<dxo_f64x2*>(S);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int k = 0; k < LPC; ++k) S2[slot(k * DXO_WAVE + lane)] = r[c][k];
            op_fence();
            if (lane / TR_PC == c) {
#pragma unroll
                for (int j = 0; j < CV; ++j) row[j] = S2[(lane - c * TR_PC) * RS + j];
            }
            op_fence();
        }
    }
};

// ---- the tangent's action from the RETURNED STATE of the von Mises operator instead of from its d x d block (dxo_tangent_apply_vm,
// dxo_tangent_diagonal_vm): a point's (sigma, dp) are 56 bytes at d = 6 where its tangent is 288, the tangent is a function of
// them (vm_tangent_state, vm_core.h — the formulas dxo_vm_expand_tangent rebuilds blocks with), and t = C e costs ~40 flops without
// the matrix (vm_tangent_times). A matrix-free Newton-Krylov solve then never needs C_tang to exist: the fused operator runs with
// C_tang = NULL (160 instead of 448 bytes per point) and every Krylov matvec reads a fifth of the bytes.
struct VmStateSrc {
    VmConst c;
    const double* sigma;   // [n][d]
    const double* dp;      // [n]
};

template <int D>
struct VmPoint {
    dxo_f64x2 s2[D / 2];
    double dp;
    __device__ __forceinline__ void request(const VmStateSrc& vs, int64_t point, bool has) {
        const dxo_f64x2* Sp = reinterpret_cast<const dxo_f64x2*>(vs.sigma + point * D);
#pragma unroll
        for (int k = 0; k < D / 2; ++k) s2[k] = has ? Sp[k] : dxo_f64x2{1.0, 0.0};     // a harmless non-singular state for idle lanes
        dp = has ? vs.dp[point] : 0.0;
    }
    __device__ __forceinline__ void state(const VmStateSrc& vs, double (&nrm)[D], double& a, double& b) const {
        double sig[D];
#pragma unroll
        for (int k = 0; k < D / 2; ++k) { sig[2 * k] = s2[k].x; sig[2 * k + 1] = s2[k].y; }
        vm_tangent_state<D>(vs.c, sig, dp, nrm, a, b);
    }
};

// Mandel / row-major operand value -> dual tensor Ghat[i][j] = d(pairing)/d(grad u)_ij, and the value part
template <int G, int BS, int KIND>
__device__ __forceinline__ void dual_tensor(const double (&s)[OperandShape<G, BS, KIND>::D], double (&vh)[BS],
                                            double (&gh)[BS][G]) {
    constexpr double r2 = 0.70710678118654752440;
#pragma unroll
    for (int i = 0; i < BS; ++i) {
        vh[i] = 0.0;
#pragma unroll
        for (int j = 0; j < G; ++j) gh[i][j] = 0.0;
    }
    if constexpr (KIND == DXO_OPERAND_VALUE) {
#pragma unroll
        for (int i = 0; i < BS; ++i) vh[i] = s[i];
    } else if constexpr (KIND == DXO_OPERAND_GRAD || KIND == DXO_OPERAND_DEFGRAD) {
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) gh[i][j] = s[i * G + j];
    } else if constexpr (KIND == DXO_OPERAND_VALUE_GRAD) {
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            vh[i] = s[i];
#pragma unroll
            for (int j = 0; j < G; ++j) gh[i][j] = s[BS + i * G + j];
        }
    } else if constexpr (KIND == DXO_OPERAND_DIV) {      // pairing s div v: Ghat = s I
#pragma unroll
        for (int i = 0; i < G; ++i) gh[i % BS][i] = s[0];
    } else {   // EPS_MANDEL: e = [g00, g11, (g22 | 0), r(g01+g10), r(g02+g20), r(g12+g21)]
        if constexpr (G == 2) {
            gh[0][0] = s[0]; gh[1][1] = s[1]; gh[0][1] = gh[1][0] = r2 * s[3];
        } else {
            gh[0][0] = s[0]; gh[1][1] = s[1]; gh[2][2] = s[2];
            gh[0][1] = gh[1][0] = r2 * s[3]; gh[0][2] = gh[2][0] = r2 * s[4]; gh[1][2] = gh[2][1] = r2 * s[5];
        }
    }
}

// Phase 1 tail + phase 2: park (vh, T) of this lane's point, then scatter the cell-node sums. W layout behind the
// gathered data: Tm[point][BS*(G+1)].
template <int G, int BS>
__device__ __forceinline__ void adjoint_scatter(const OperandDev& m, const double* tab, double* Tm, bool active, int lane,
                                                const double (&vh)[BS], const double (&gh)[BS][G], const double (&K)[G][G],
                                                double scale, int64_t c0, int ncell, const int32_t* __restrict__ cells,
                                                double* __restrict__ out, double* __restrict__ fe) {
    const OperandLayout<G> L(m);
    // per-point stride: an ODD number of doubles. Phase 2's lanes read the same slot of up to three different cells at once
    // (cell stride = nq points); with the natural stride 12 (hexahedra: 8 x 12 doubles = 192 dwords = 0 mod 64 banks) those
    // reads collided on one bank — a quarter of the kernel's LDS cycles were conflict cycles (profiles/r04_device_loop_sq.json)
    constexpr int PT = DXO_ADJ_PAD ? ((BS * (G + 1)) | 1) : BS * (G + 1);
    if (active) {
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            Tm[lane * PT + i] = scale * vh[i];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < G; ++j) t += gh[i][j] * K[k][j];
                Tm[lane * PT + BS + i * G + k] = scale * t;
            }
        }
    }
    op_fence();
    const int nd = m.ndofs, nq = m.nq;
    for (int idx = lane; idx < ncell * nd; idx += DXO_WAVE) {
        const int c = idx / nd, a = idx - c * nd;
        double acc[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) acc[i] = 0.0;
        for (int q = 0; q < nq; ++q) {
            const double* T = Tm + (c * nq + q) * PT;
            const double ph = tab[q * L.sphi + a];
            const double* dp = tab + L.o_dphi + q * L.sdphi + a * G;
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                double t = T[i] * ph;
#pragma unroll
                for (int k = 0; k < G; ++k) t += T[BS + i * G + k] * dp[k];
                acc[i] += t;
            }
        }
        const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
        if (fe) {        // two-pass form: the element vector entry, summed per node afterwards (node_sum).
            // Layout fe[a][cell][i]: the cells of a wave group are consecutive, so each local node's entries leave as one
            // contiguous run per group, and in node_sum neighbouring nodes (same local role in neighbouring cells) read
            // neighbouring addresses.
#pragma unroll
            for (int i = 0; i < BS; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * BS + i] = acc[i];
        } else {
            const int64_t node = m.dofmap[cell * nd + a];
#pragma unroll
            for (int i = 0; i < BS; ++i) unsafeAtomicAdd(out + node * BS + i, acc[i]);
        }
    }
    op_fence();
}

template <int G, int BS, int KIND>
__global__ __launch_bounds__(DXO_BLOCK) void operand_adjoint(OperandDev m, const double* __restrict__ wq, int lds_wave,
                                                             const double* __restrict__ S, const int32_t* __restrict__ cells,
                                                             int64_t n_cells, double* __restrict__ out,
                                                             double* __restrict__ fe) {
    constexpr int D = OperandShape<G, BS, KIND>::D;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * lds_wave;
    const int cpw = m.cells_per_wave;
    double* Tm = W + cpw * (op_odd(m.ndofs * BS) + op_odd(m.ngeom * G));
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = (n_cells - c0 < cpw) ? (int)(n_cells - c0) : cpw;
        // only the geometry is needed
        {
            const int ng = m.ngeom, sx = op_odd(ng * G);
            double* X = W + cpw * op_odd(m.ndofs * BS);
            for (int idx = lane; idx < ncell * ng; idx += DXO_WAVE) {
                const int c = idx / ng, v = idx - c * ng;
                const int64_t cell = cells ? (int64_t)cells[c0 + c] : c0 + c;
                const int64_t node = m.geom_dofmap[cell * ng + v];
#pragma unroll
                for (int j = 0; j < G; ++j) X[c * sx + v * G + j] = m.x[node * G + j];
            }
        }
        op_fence();
        const int c = lane / m.nq, q = lane - c * m.nq;
        const bool active = c < ncell;
        double K[G][G], vh[BS], gh[BS][G], scale = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) K[i][j] = 0.0;
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            vh[i] = 0.0;
#pragma unroll
            for (int j = 0; j < G; ++j) gh[i][j] = 0.0;
        }
        if (active) {
            const OperandLayout<G> L(m);
            const double* dpsi = tab + L.o_dpsi + q * L.sdpsi;
            const double* Xc = W + cpw * op_odd(m.ndofs * BS) + c * L.sx;
            double J[G][G];
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] = 0.0;
            for (int v = 0; v < m.ngeom; ++v)
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int k = 0; k < G; ++k) J[j][k] += Xc[v * G + j] * dpsi[v * G + k];
            const double det = invert<G>(J, K);
            scale = wq[q] * fabs(det);
            double s[D];
            const double* Sp = S + ((c0 + c) * m.nq + q) * D;
#pragma unroll
            for (int k = 0; k < D; ++k) s[k] = Sp[k];
            dual_tensor<G, BS, KIND>(s, vh, gh);
        }
        adjoint_scatter<G, BS>(m, tab, Tm, active, lane, vh, gh, K, scale, c0, ncell, cells, out, fe);
    }
}

#ifndef DXO_NS_WIDE
#define DXO_NS_WIDE 1    // node_sum: 16-byte index and element-vector loads
#endif
#ifndef DXO_C8_ADJ_BLOCKS_PER_CU
#define DXO_C8_ADJ_BLOCKS_PER_CU 32   // operand_adjoint_c8 grid; 4 / 8 / 16 / 32 / 64 / uncapped workgroups per CU: 0.780 / 0.754 / 0.742 / 0.718 / 0.725 / 0.818 ms per call
#endif
#ifndef DXO_TD_VM_BLOCKS_PER_CU
#define DXO_TD_VM_BLOCKS_PER_CU 8    // tangent_diag<..., VM>: 4 / 8 / 16 make no difference (0.999-1.013 ms hexahedra, 0.339-0.346 triangles)
#endif
#ifndef DXO_TD_BLOCKS_PER_CU
#define DXO_TD_BLOCKS_PER_CU 8
#endif
#ifndef DXO_TA_BLOCKS_PER_CU
#define DXO_TA_BLOCKS_PER_CU 8
#endif
#ifndef DXO_NS_BLOCKS_PER_CU
#define DXO_NS_BLOCKS_PER_CU 1024   // node_sum grid: in effect one thread per node. 4 / 8 / 16 / 32 / 64 / uncapped workgroups per CU, state-based matvec:
                                    // hexahedra 0.959 / 0.956 / 0.936 / 0.929 / 0.948 / 0.922 ms, triangles 0.395 / 0.402 / 0.400 / 0.402 / 0.395 / 0.388
#endif
#ifndef DXO_TA_VM_BLOCKS_PER_CU
#define DXO_TA_VM_BLOCKS_PER_CU 12     // persistent grid of tangent_apply<..., VM>; 2 / 4 / 8 / 12 / 24 / 48 workgroups per CU: hexahedra 0.954 / 0.970 / 0.975 / 0.956 / 0.999 / 1.054 ms, triangles 0.459 / 0.460 / 0.422 / 0.415 / 0.429 / 0.457
#endif
#ifndef DXO_TA_VM_UT
#define DXO_TA_VM_UT 1      // node groups of the register scatter in flight in tangent_apply<..., VM>
#endif
#ifndef DXO_C8_ADJ_UT
#define DXO_C8_ADJ_UT 1     // the same in operand_adjoint_c8
#endif
typedef uint32_t NodeEnt4 __attribute__((ext_vector_type(4), aligned(4)));
typedef double NodeF64x2 __attribute__((ext_vector_type(2), aligned(8)));

// second pass of the two-pass form: out[node] += sum of the node's element-vector entries, in the fixed order of the
// transposed dofmap — no atomics, bit-reproducible
template <int BS>
__global__ __launch_bounds__(DXO_BLOCK) void node_sum(int64_t n_nodes, const int64_t* __restrict__ ptr,
                                                      const uint32_t* __restrict__ ent, const double* __restrict__ fe,
                                                      double* __restrict__ out, int overwrite) {
    // A node's sum is a chain of dependent loads (ptr -> ent -> fe) per entry: with one entry at a time the wave sat in s_waitcnt
    // 92 % of its cycles (profiles/r04_device_loop_pmc.json). The entries are taken four at a time — the four indices, then the
    // four element-vector pieces, are in flight together — and `out` is requested before the loop. The additions keep the order
    // of the transposed dofmap (bit-reproducible, same result as the one-at-a-time form).
    constexpr int U = 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += stride) {
        const int64_t e0 = ptr[n], e1 = ptr[n + 1];
        double acc[BS], cur[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            acc[i] = 0.0;
            cur[i] = overwrite ? 0.0 : out[n * BS + i];     // option consumer_overwrite: out = sum (no memset before, no read here)
        }
        for (int64_t e = e0; e < e1; e += U) {
            uint32_t idx[U];
            double f[U][BS];
            // the kernel is bound by the ISSUE of its scattered loads (SQ_WAIT_INST_ANY 0.49 of the wave cycles): the four indices are
            // one 16-byte load (4-byte aligned: the array is padded by U - 1 entries), an entry's BS doubles one or two loads
#if DXO_NS_WIDE
            const NodeEnt4 i4 = *reinterpret_cast<const NodeEnt4*>(ent + e);
#pragma unroll
            for (int k = 0; k < U; ++k) idx[k] = e + k < e1 ? i4[k] : 0u;
#else
#pragma unroll
            for (int k = 0; k < U; ++k) idx[k] = e + k < e1 ? ent[e + k] : 0u;
#endif
#pragma unroll
            for (int k = 0; k < U; ++k) {
                const double* src = fe + (int64_t)idx[k] * BS;
                if constexpr (BS >= 2 && DXO_NS_WIDE) {
                    const NodeF64x2 v = *reinterpret_cast<const NodeF64x2*>(src);
                    f[k][0] = v.x;
                    f[k][1] = v.y;
                    if constexpr (BS == 3) f[k][2] = src[2];
                } else {
#pragma unroll
                    for (int i = 0; i < BS; ++i) f[k][i] = src[i];
                }
                if (!(e + k < e1)) {
#pragma unroll
                    for (int i = 0; i < BS; ++i) f[k][i] = 0.0;
                }
            }
#pragma unroll
            for (int k = 0; k < U; ++k)
                if (e + k < e1) {
#pragma unroll
                    for (int i = 0; i < BS; ++i) acc[i] += f[k][i];
                }
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) out[n * BS + i] = cur[i] + acc[i];
    }
}

// Internal force on cells of eight points and at most 32 nodes (Q2 / Q1 hexahedra, 2x2x2 rule), kind EPS_MANDEL: no LDS staging at
// all. Lane = (cell, point): vertex q of the cell goes from global memory into lane q's registers (requested one group ahead) and
// the 8 lanes all-gather the coordinates for J (cell8_dpp.h); the point's stress row is three 16-byte loads at a 48-byte lane
// stride (every line is used in full by consecutive lanes); the pulled-back tensor stays in registers and the element-vector
// entries are reduce-scattered over the cell's lanes. ~100 registers. Replaces the lane = cell kernel (adjoint_cell_eps: the
// whole cell in one lane's registers, one wave per SIMD, 0.56 ms per 10^7 points) on these elements.
template <int ND>
__global__ __launch_bounds__(DXO_BLOCK) void operand_adjoint_c8(OperandDev m, const double* __restrict__ wq, const double* __restrict__ S,
                                                                int64_t n_cells, double* __restrict__ out, double* __restrict__ fe) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    c8_fill_tables(m, lds);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const C8Lane L(lds, lane);
    constexpr int cpw = 8;
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    const double w_l = wq[lane & 7];
    const int c_l = lane >> 3, q_l = lane & 7;
    auto vertex_index = [&](int64_t g) -> int32_t { return c_l < cells_in(g) ? m.geom_dofmap[(g * cpw + c_l) * 8 + q_l] : -1; };
    auto vertex = [&](int32_t xn, double (&xv)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[j] = xn >= 0 ? m.x[(int64_t)xn * 3 + j] : 0.0;
    };
    int64_t grp = walk.first;
    int32_t xn = vertex_index(grp);
    double xv[3];
    vertex(xn, xv);
    xn = vertex_index(grp + stride);
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = cells_in(grp);
        const bool has_point = c_l < ncell;
        dxo_f64x2 s2[3];
        {
            const dxo_f64x2* Sp = reinterpret_cast<const dxo_f64x2*>(S + (c0 * 8 + lane) * 6);
#pragma unroll
            for (int k = 0; k < 3; ++k) s2[k] = has_point ? Sp[k] : dxo_f64x2{0.0, 0.0};
        }
        double K[3][3];
        const double det = c8_geometry(L, xv, K);
        vertex(xn, xv);                              // the next group's vertex (its index has been here for an iteration)
        xn = vertex_index(grp + 2 * stride);
        const double s[6] = {s2[0].x, s2[0].y, s2[1].x, s2[1].y, s2[2].x, s2[2].y};
        double vh[3], gh[3][3], T[3][3];
        dual_tensor<3, 3, DXO_OPERAND_EPS_MANDEL>(s, vh, gh);
        const double scale = w_l * fabs(det);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double tt = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) tt += gh[i][j] * K[k][j];
                T[i][k] = has_point ? scale * tt : 0.0;       // lanes without a point: zero vertices, singular J
            }
        const int64_t cell = c0 + c_l;
        c8_scatter<ND, DXO_C8_ADJ_UT>(L, T, [&](int a, const double (&o)[3]) {
            if (!has_point) return;
            if (fe) {
#pragma unroll
                for (int i = 0; i < 3; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * 3 + i] = o[i];
            } else {
                const int64_t node = m.dofmap[cell * ND + a];
#pragma unroll
                for (int i = 0; i < 3; ++i) unsafeAtomicAdd(out + node * 3 + i, o[i]);
            }
        });
    }
}

// operand_adjoint_c8 with the element-vector contraction on the f64 matrix pipe (cell8_mfma.h; option adjoint_mfma)
#if defined(DXO_EXPERIMENTS) && defined(DXO_C8M_FORWARD) && DXO_C8M_FORWARD
#define DXO_C8M_FWD 1
#include "../../scripts/exp/adjoint_mfma_forward.h"      // c8m_forward_eps: the strain contraction as MFMAs too (measured, not shipped)
#else
#define DXO_C8M_FWD 0
constexpr int C8M_FTAB = 0;
#endif

template <int ND>
__global__ __launch_bounds__(DXO_BLOCK) void operand_adjoint_c8_mfma(OperandDev m, const double* __restrict__ wq, const double* __restrict__ S,
                                                                     int64_t n_cells, double* __restrict__ out, double* __restrict__ fe) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    // geometry rows only (the dphi table lives in the A fragments): [q][vertex][dpsi_x, dpsi_y, dpsi_z, 0] at the offset C8Lane expects
    for (int e = threadIdx.x; e < C8_GEO; e += blockDim.x) {
        const int k = e & 3, v = (e >> 2) & 7, q = e >> 5;
        lds[C8_TAB + e] = k < 3 ? m.dpsi[(q * 8 + v) * 3 + k] : 0.0;
    }
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* Tl = lds + C8_LDS + wave * C8M_WAVE;
    __syncthreads();
    const C8Lane L(lds, lane);
    constexpr int cpw = 8;
    double Afr[2][6];
    c8m_load_A<ND>(m, lane, Afr);
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    const double w_l = wq[lane & 7];
    const int c_l = lane >> 3, q_l = lane & 7;
    auto vertex_index = [&](int64_t g) -> int32_t { return c_l < cells_in(g) ? m.geom_dofmap[(g * cpw + c_l) * 8 + q_l] : -1; };
    auto vertex = [&](int32_t xn, double (&xv)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[j] = xn >= 0 ? m.x[(int64_t)xn * 3 + j] : 0.0;
    };
    int64_t grp = walk.first;
    int32_t xn = vertex_index(grp);
    double xv[3];
    vertex(xn, xv);
    xn = vertex_index(grp + stride);
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = cells_in(grp);
        const bool has_point = c_l < ncell;
        dxo_f64x2 s2[3];
        {
            const dxo_f64x2* Sp = reinterpret_cast<const dxo_f64x2*>(S + (c0 * 8 + lane) * 6);
#pragma unroll
            for (int k = 0; k < 3; ++k) s2[k] = has_point ? Sp[k] : dxo_f64x2{0.0, 0.0};
        }
        double K[3][3];
        const double det = c8_geometry(L, xv, K);
        vertex(xn, xv);                              // the next group's vertex (its index has been here for an iteration)
        xn = vertex_index(grp + 2 * stride);
        const double s[6] = {s2[0].x, s2[0].y, s2[1].x, s2[1].y, s2[2].x, s2[2].y};
        double vh[3], gh[3][3];
        dual_tensor<3, 3, DXO_OPERAND_EPS_MANDEL>(s, vh, gh);
        const double scale = w_l * fabs(det);
        double T[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double tt = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) tt += gh[i][j] * K[k][j];
                T[i][k] = has_point ? scale * tt : 0.0;       // lanes without a point: zero vertices, singular J
            }
        c8m_d4 acc[2][2];
        c8m_contract(Tl, lane, T, Afr, acc);
        c8m_store<ND>(m, lane, acc, c0, ncell, fe, out);
    }
}

#ifdef DXO_EXPERIMENTS      // operand_adjoint_c8_patch: the patch form (measured slower, profiles/r05_patch_form.txt)
#define DXO_PATCH_PART 1
#include "../../scripts/exp/adjoint_patch_kernels.h"
#undef DXO_PATCH_PART
#endif

// K v without K: gather v, eps(v) per point, t = C_tang e, scatter B^T t (bs = gdim, Mandel pairing).
// Round 4: the kernel spent two thirds of its wave cycles parked in s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.65) — the
// dofmap -> v gather was a dependent pair of scattered loads issued and awaited inside every group, and the point's 36
// tangent entries were requested only after the strain was known. Now (i) the gather runs as the register pipeline vm_field
// uses (values of group g+1 and indices of group g+2 in flight while group g is computed), (ii) the lane's tangent row is
// requested at the top of the iteration, 16 bytes per load, and is consumed after the contraction has hidden its latency,
// (iii) the parked tensors have an odd stride (adjoint_scatter). ND_CT / NG_CT as in operand_compute_geo.
// VM: the tangent's action comes from the von Mises state (VmStateSrc) instead of from C_tang rows.
// MF (option adjoint_mfma, Q2 hexahedra): the scatter's contraction as f64 MFMAs (c8m_contract) instead of the DPP reduce-scatter.
template <int G, int ND_CT = 0, int NG_CT = 0, bool VM = false, bool MF = false>
__global__ __launch_bounds__(DXO_BLOCK, (MF && ND_CT != 27) ? DXO_TA_GM_WAVES : VM ? (MF ? DXO_TA_VM_MF_WAVES : DXO_TA_VM_WAVES) : DXO_TA_WAVES) void tangent_apply(OperandDev m, const double* __restrict__ wq, int lds_wave,
                                                              const double* __restrict__ C_tang, VmStateSrc vs,
                                                              const double* __restrict__ v, int64_t n_cells,
                                                              double* __restrict__ out, double* __restrict__ fe) {
    constexpr int D = G == 2 ? 4 : 6;
    constexpr int CV = D * D / 2;        // 16-byte pieces of a point's tangent
    // cells of 8 points and at most 32 nodes (launched so only for nq = 8): the scatter phase runs in registers, a DPP
    // reduce-scatter over the cell's 8 lanes (cell8_dpp.h) instead of parked tensors and 96 LDS reads per (cell, node) pair
    constexpr bool RS = DXO_TA_RS && G == 3 && ND_CT > 0 && ND_CT <= C8_NODES && NG_CT == 8;
    // MF on the other standard elements (scatter_mfma.h): P2 tetrahedra with the 4-point rule, P2 triangles with the 3-point rule
    constexpr bool GMF = MF && !RS && ND_CT > 0;
    constexpr int NQ_GM = G == 3 ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    double* tabP = lds + m.table_doubles + (DXO_BLOCK / DXO_WAVE) * lds_wave;     // behind the waves' regions (RS, GMF)
    if constexpr (GMF) gm_fill_A<G, ND_CT, NQ_GM>(m, tabP);
    if constexpr (RS && !MF) c8_fill_tables(m, tabP);
    if constexpr (RS && MF) c8m_fill_A<ND_CT>(m, tabP);      // 768 of the C8_LDS doubles
#if DXO_C8M_FWD
    if constexpr (RS && MF) c8m_fill_F<ND_CT>(m, tabP + 12 * DXO_WAVE);
#endif
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * lds_wave;
    const int cpw = m.cells_per_wave;
    double* Tm = W + cpw * (op_odd(m.ndofs * G) + op_odd(m.ngeom * G));
    const C8Lane L8(tabP, lane);
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    const bool piped = DXO_TA_PIPE && operand_can_pipe(m);
    OperandPipe<G, G> pf;
    int64_t grp = walk.first;
    if (piped) {
        pipe_load_indices<G, G>(m, pf, grp * cpw, cells_in(grp), lane);
        pipe_load_values<G, G>(m, pf, v);
        pipe_load_indices<G, G>(m, pf, (grp + stride) * cpw, cells_in(grp + stride), lane);
    }
    const int q_l = lane - (lane / m.nq) * m.nq;
    const double w_l = lane < cpw * m.nq ? wq[q_l] : 0.0;
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = cells_in(grp);
        const bool has_point = lane < ncell * m.nq;
        // the tangent of this lane's point: requested now, used after the contraction
#if DXO_TA_STAGE
        TangentRows<D> rows;
        VmPoint<D> vp;
        if constexpr (VM) vp.request(vs, c0 * m.nq + lane, has_point);
#if DXO_TA_EARLY_C
        if constexpr (!VM) rows.request(C_tang, c0 * m.nq, ncell * m.nq, lane);
#endif
#else
        dxo_f64x2 Cq[CV];
        const dxo_f64x2* Cp = reinterpret_cast<const dxo_f64x2*>(C_tang + (c0 * m.nq + lane) * (D * D));
#if DXO_TA_EARLY_C
#pragma unroll
        for (int k = 0; k < CV; ++k) Cq[k] = has_point ? ta_load<DXO_TA_NT != 0>(Cp + k) : dxo_f64x2{0.0, 0.0};
#endif
#endif
        if (piped) {
            pipe_commit<G, G>(m, pf, W, ncell, lane);
            pipe_load_values<G, G>(m, pf, v);
            pipe_load_indices<G, G>(m, pf, (grp + 2 * stride) * cpw, cells_in(grp + 2 * stride), lane);
        } else {
            operand_gather<G, G>(m, W, v, nullptr, c0, ncell, lane);
        }
        double e[D], K[G][G], det = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) K[i][j] = 0.0;
        bool active;
#if DXO_C8M_FWD
        if constexpr (RS && MF) active = c8m_forward_eps<ND_CT>(m, tab, tabP + 12 * DXO_WAVE, W, ncell, lane, e, K, det);
        else
#endif
        active = operand_compute_geo<G, G, DXO_OPERAND_EPS_MANDEL, ND_CT, NG_CT>(m, tab, W, ncell, lane, e, K, det);
        double vh[G], gh[G][G], scale = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            vh[i] = 0.0;
#pragma unroll
            for (int j = 0; j < G; ++j) gh[i][j] = 0.0;
        }
        double t[D];
#if DXO_TA_STAGE
        if (!active) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[k] = 0.0;
        }
        if constexpr (VM) {
            double nrm[D], a, b;
            vp.state(vs, nrm, a, b);
            vm_tangent_times<D>(vs.c, nrm, a, b, e, t);
        } else {
#if !DXO_TA_EARLY_C
            rows.request(C_tang, c0 * m.nq, ncell * m.nq, lane);
#endif
            rows.times(W, lane, e, t);     // compute_geo has fenced: the gather buffer is free, the parked tensors are not written yet
        }
        if (active) {
            scale = w_l * fabs(det);
            dual_tensor<G, G, DXO_OPERAND_EPS_MANDEL>(t, vh, gh);
        }
#else
        if (active) {
            scale = w_l * fabs(det);
#if !DXO_TA_EARLY_C
#pragma unroll
            for (int k = 0; k < CV; ++k) Cq[k] = ta_load<DXO_TA_NT != 0>(Cp + k);
#endif
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
            dual_tensor<G, G, DXO_OPERAND_EPS_MANDEL>(t, vh, gh);
        }
#endif
        if constexpr (RS) {
            double T[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) tt += gh[i][j] * K[k][j];
                    T[i][k] = scale * tt;          // scale = 0 and gh = 0 for lanes without a point
                }
            if constexpr (MF) {
                c8m_d4 acc[2][2];
                const double none[2][6] = {};
                c8m_contract<true>(W, lane, T, none, acc, tabP);       // compute_geo / rows.times have fenced: the gather buffer is free
                c8m_store<ND_CT>(m, lane, acc, c0, ncell, fe, out);
                continue;
            }
            const int64_t cell = c0 + (lane >> 3);
            c8_scatter<ND_CT, VM ? DXO_TA_VM_UT : 1>(L8, T, [&](int a, const double (&o)[3]) {
                if (!active) return;
                if (fe) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * 3 + i] = o[i];
                } else {
                    const int64_t node = m.dofmap[cell * ND_CT + a];
#pragma unroll
                    for (int i = 0; i < 3; ++i) unsafeAtomicAdd(out + node * 3 + i, o[i]);
                }
            });
        } else if constexpr (GMF) {
            double T[G][G];
#pragma unroll
            for (int i = 0; i < G; ++i)
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < G; ++j) tt += gh[i][j] * K[k][j];
                    T[i][k] = scale * tt;          // scale = 0 and gh = 0 for lanes without a point
                }
            gm_scatter<G, ND_CT, NQ_GM>(m, tabP, W, lane, T, c0, ncell, fe, out);      // compute_geo / rows.times have fenced: the wave's region is free
        } else {
            adjoint_scatter<G, G>(m, tab, Tm, active, lane, vh, gh, K, scale, c0, ncell, nullptr, out, fe);
        }
    }
}

#ifdef DXO_EXPERIMENTS
#include "../../scripts/exp/adjoint_variants.h"      // tangent_apply_c8 (measured, not shipped)
#endif

// diag(K) without K, for a Jacobi preconditioner: K_(a,i),(a,i) = sum_q w|detJ| e_(a,i)^T C_q e_(a,i) with e_(a,i) the Mandel
// strain of the unit dof (local node a, component i). With g = K^T dphi_a (physical gradient of the node's basis function) that
// strain is linear in g, e = E_i g, so the quadratic form is g^T M_i g with the 3x3 (2x2) matrix M_i = E_i^T C E_i picked out of C
// (rows / columns m(i,j) of the Mandel vector, weight 1 on the diagonal pair and 1/sqrt2 otherwise), and in reference gradients
// dphi^T (w|detJ| K M_i K^T) dphi. Round 4: phase 1, lane = (cell, point), builds the SYMMETRISED reference-space matrices
// NS_i (G(G+1)/2 numbers per component: 18 per point on hexahedra instead of the 36 of C) from its own tangent row, requested
// at the top of the iteration, and parks them in the wave's LDS slice; phase 2, lane = (cell, node), sums
// dphi_k dphi_k' NS_i[kk'] over the cell's points. Before, phase 2 read the 36 entries of C_q from global memory for every
// (node, point) pair — 27 times each on Q2 hexahedra: 4.5 ms per 10^7 points, 2.3 matvecs' worth; now see profiles/README.md.
// ND_CT > 0 (launched for cells of 8 points, at most 32 nodes): phase 2 in registers with the DPP reduce-scatter of cell8_dpp.h.
template <int G, int ND_CT = 0, bool VM = false, bool MF = false>
__global__ __launch_bounds__(DXO_BLOCK, 2) void tangent_diag(OperandDev m, const double* __restrict__ wq, int lds_wave,
                                                            const double* __restrict__ C_tang, VmStateSrc vs, int64_t n_cells,
                                                            double* __restrict__ out, double* __restrict__ fe) {
    constexpr int D = G == 2 ? 4 : 6;
    constexpr int CV = D * D / 2;
    constexpr int NS1 = G * (G + 1) / 2;            // unique entries of one symmetrised matrix
    constexpr int PT = (G * NS1) | 1;               // odd per-point stride (bank spread across cells)
    constexpr double r2 = 0.70710678118654752440;
    constexpr bool RS = DXO_TA_RS && G == 3 && (ND_CT == 27 || ND_CT == 8);
    // MF on P2 tetrahedra (4-point rule) / P2 triangles (3-point rule): scatter_mfma.h, rows (q, pair) in passes of three pairs
    constexpr bool GMF = MF && !RS && ND_CT > 0;
    constexpr int NQ_GM = G == 3 ? 4 : 3;
    using GS = GmShape<G, GMF ? ND_CT : 1, NQ_GM, 3>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    double* tabP = lds + m.table_doubles + (DXO_BLOCK / DXO_WAVE) * lds_wave;
    if constexpr (RS && !MF) c8_fill_tables(m, tabP);
    if constexpr (RS && MF) c8m_fill_A2<ND_CT>(m, tabP);
    if constexpr (GMF) gm_fill_A2<G, ND_CT, NQ_GM>(m, tabP);
    __syncthreads();
    const OperandLayout<G> L(m);
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * lds_wave;
    const C8Lane L8(tabP, lane);
    const int cpw = m.cells_per_wave, nd = m.ndofs, nq = m.nq, ng = m.ngeom;
    const int sx = op_odd(ng * G);
    double* X = W;
    double* Pm = X + ((cpw * sx + 1) & ~1);          // [point][PT]; even offset: the slice also stages the tangent rows (16-byte units)
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    const int c_l = lane / nq, q_l = lane - c_l * nq;
    const double w_l = lane < cpw * nq ? wq[q_l] : 0.0;
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = (n_cells - c0 < cpw) ? (int)(n_cells - c0) : cpw;
        const bool has_point = c_l < ncell;
        dxo_f64x2 Cq[CV];
        double vn[D], va = 0.0, vb = 0.0;       // VM: the point's tangent state
#if DXO_TA_STAGE
        TangentRows<D> rows;
        if constexpr (VM) {
            VmPoint<D> vp;
            vp.request(vs, c0 * nq + lane, has_point);
            vp.state(vs, vn, va, vb);
        } else {
            rows.request(C_tang, c0 * nq, ncell * nq, lane);
            rows.deliver(Pm, lane, Cq);          // the parked matrices of the last group have been consumed (fence at the loop's end)
        }
#else
        {
            const dxo_f64x2* Cp = reinterpret_cast<const dxo_f64x2*>(C_tang + (c0 * nq + lane) * (D * D));
#pragma unroll
            for (int k = 0; k < CV; ++k) Cq[k] = has_point ? ta_load<DXO_TA_NT != 0>(Cp + k) : dxo_f64x2{0.0, 0.0};
        }
#endif
        for (int idx = lane; idx < ncell * ng; idx += DXO_WAVE) {
            const int c = idx / ng, v = idx - c * ng;
            const int64_t node = m.geom_dofmap[(c0 + c) * ng + v];
#pragma unroll
            for (int j = 0; j < G; ++j) X[c * sx + v * G + j] = m.x[node * G + j];
        }
        op_fence();
        double NSr[G * NS1];             // RS: this point's matrices stay in registers
#pragma unroll
        for (int k = 0; k < G * NS1; ++k) NSr[k] = 0.0;
        if (has_point) {
            const double* dpsi = tab + L.o_dpsi + q_l * L.sdpsi;
            const double* Xc = X + c_l * sx;
            double J[G][G], K[G][G];
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] = 0.0;
            for (int v = 0; v < ng; ++v)
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int k = 0; k < G; ++k) J[j][k] += Xc[v * G + j] * dpsi[v * G + k];
            const double det = invert<G>(J, K);
            const double scale = w_l * fabs(det);
            auto Cat = [&](int r, int cc) -> double {
                if constexpr (VM) return c_elas_ij(vs.c, r, cc) - va * (vn[r] * vn[cc]) - vb * dev_ij(r, cc);
                const dxo_f64x2 c2 = Cq[(r * D + cc) / 2];
                return ((r * D + cc) & 1) ? c2.y : c2.x;
            };
#pragma unroll
            for (int i = 0; i < G; ++i) {
                // Mandel slot of the pair (i, j) and its weight: the diagonal pair sits at i with weight 1, an off-diagonal pair at
                // 3 (01), 4 (02), 5 (12) with weight 1/sqrt2 (two dimensions: only (01) -> 3)
                int mi[G];
                double wi[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    mi[j] = i == j ? i : (G == 2 ? 3 : (i + j + 2));
                    wi[j] = i == j ? 1.0 : r2;
                }
                double M[G][G], KM[G][G];
#pragma unroll
                for (int j = 0; j < G; ++j)
#pragma unroll
                    for (int jj = 0; jj < G; ++jj) M[j][jj] = wi[j] * wi[jj] * Cat(mi[j], mi[jj]);
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int jj = 0; jj < G; ++jj) {
                        double t = 0.0;
#pragma unroll
                        for (int j = 0; j < G; ++j) t += K[k][j] * M[j][jj];
                        KM[k][jj] = t;
                    }
                int slot = 0;
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int kk = k; kk < G; ++kk) {
                        double a = 0.0, b = 0.0;
#pragma unroll
                        for (int jj = 0; jj < G; ++jj) {
                            a += KM[k][jj] * K[kk][jj];
                            b += KM[kk][jj] * K[k][jj];
                        }
                        if constexpr (RS || GMF) NSr[i * NS1 + slot] = scale * (k == kk ? a : a + b);
                        else Pm[lane * PT + i * NS1 + slot] = scale * (k == kk ? a : a + b);
                        ++slot;
                    }
            }
        }
        if constexpr (GMF) {
            c8m_d4 acc[GS::MT][GS::NT];
            double T[G][3];
#pragma unroll
            for (int p = 0; p < NS1 / 3; ++p) {
#pragma unroll
                for (int i = 0; i < G; ++i)
#pragma unroll
                    for (int k = 0; k < 3; ++k) T[i][k] = NSr[i * NS1 + 3 * p + k];
                if (p == 0) gm_contract<G, ND_CT, NQ_GM, 3, false>(tabP, Pm, lane, T, acc);
                else gm_contract<G, ND_CT, NQ_GM, 3, true>(tabP + GS::ATAB, Pm, lane, T, acc);
            }
            gm_store<G, ND_CT, NQ_GM>(m, lane, acc, c0, ncell, fe, out);
            continue;      // gm_contract has fenced: the staging slice (Pm) is free for the next group
        }
        if constexpr (RS) {
            // phase 2 in registers: every lane forms its point's partial of K_(a,i),(a,i) for 8 nodes at a time and the cell's 8
            // lanes reduce-scatter them (cell8_dpp.h); the lane ends up with the entries of its own four nodes
            if constexpr (MF) {
                // the same sums on the matrix pipe: rows (q, pair) of the two passes against the product tables (c8m_fill_A2)
                c8m_d4 acc[2][2];
                const double none[2][6] = {};
                double T[3][3];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int k = 0; k < 3; ++k) T[i][k] = NSr[i * NS1 + 3 * p + k];
                    if (p == 0) c8m_contract<true, false>(Pm, lane, T, none, acc, tabP);
                    else c8m_contract<true, true>(Pm, lane, T, none, acc, tabP + 12 * DXO_WAVE);
                }
                c8m_store<ND_CT>(m, lane, acc, c0, ncell, fe, out);
                continue;      // c8m_contract has fenced: the staging slice (Pm) is free for the next group's tangent rows
            }
            const int64_t cell = c0 + (lane >> 3);
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                double pp[8][NS1];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double* r = L8.row(j, t);
                    const dxo_f64x2 a = *reinterpret_cast<const dxo_f64x2*>(r);
                    const double d[3] = {a.x, a.y, r[2]};
                    int slot = 0;
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int kk = k; kk < 3; ++kk) pp[j][slot++] = d[k] * d[kk];
                }
                double o[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double pj[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        double tt = 0.0;
#pragma unroll
                        for (int s2 = 0; s2 < NS1; ++s2) tt += pp[j][s2] * NSr[i * NS1 + s2];
                        pj[j] = tt;
                    }
                    o[i] = c8_reduce_scatter(pj);
                }
                const int a = L8.node0 + t;
                if (!has_point || a >= ND_CT) continue;
                if (fe) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * 3 + i] = o[i];
                } else {
                    const int64_t node = m.dofmap[cell * ND_CT + a];
#pragma unroll
                    for (int i = 0; i < 3; ++i) unsafeAtomicAdd(out + node * 3 + i, o[i]);
                }
            }
            op_fence();      // the staging slice (Pm) is rewritten by the next group's tangent rows
            continue;
        }
        op_fence();
        for (int idx = lane; idx < ncell * nd; idx += DXO_WAVE) {
            const int c = idx / nd, a = idx - c * nd;
            double acc[G];
#pragma unroll
            for (int i = 0; i < G; ++i) acc[i] = 0.0;
            for (int q = 0; q < nq; ++q) {
                const double* P = Pm + (c * nq + q) * PT;
                const double* dp = tab + L.o_dphi + q * L.sdphi + a * G;
                double d[G], pp[NS1];
#pragma unroll
                for (int k = 0; k < G; ++k) d[k] = dp[k];
                int slot = 0;
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int kk = k; kk < G; ++kk) pp[slot++] = d[k] * d[kk];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    double t = 0.0;
#pragma unroll
                    for (int s2 = 0; s2 < NS1; ++s2) t += pp[s2] * P[i * NS1 + s2];
                    acc[i] += t;
                }
            }
            const int64_t cell = c0 + c;
            if (fe) {
#pragma unroll
                for (int i = 0; i < G; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * G + i] = acc[i];
            } else {
                const int64_t node = m.dofmap[cell * nd + a];
#pragma unroll
                for (int i = 0; i < G; ++i) unsafeAtomicAdd(out + node * G + i, acc[i]);
            }
        }
        op_fence();
    }
}

int diag_lds_wave(const dxo_mesh* m) {
    const OperandDev& v = m->dev;
    const int G = m->gdim;
    const int D = G == 2 ? 4 : 6;
    int park = DXO_WAVE * ((G * (G * (G + 1) / 2)) | 1);
    if (park < TR_PC * (D * D + 2)) park = TR_PC * (D * D + 2);      // the parked-matrix slice doubles as the staging space of the tangent rows
    int wd = ((v.cells_per_wave * op_odd(v.ngeom * G) + 1) & ~1) + park;
    return (wd + 1) & ~1;
}

// tangent_apply with the register scatter parks nothing: the gather buffer, or the staging space of the tangent rows
int apply_rs_lds_wave(const dxo_mesh* m) {
    const OperandDev& v = m->dev;
    const int G = m->gdim, D = G == 2 ? 4 : 6;
    int wd = v.cells_per_wave * (op_odd(v.ndofs * G) + op_odd(v.ngeom * G));
    if (wd < TR_PC * (D * D + 2)) wd = TR_PC * (D * D + 2);
    return (wd + 1) & ~1;
}

int adjoint_lds_wave(const dxo_mesh* m) {
    const OperandDev& v = m->dev;
    const int G = m->gdim;
    int wd = v.cells_per_wave * (op_odd(v.ndofs * G) + op_odd(v.ngeom * G)) + DXO_WAVE * (DXO_ADJ_PAD ? ((G * (G + 1)) | 1) : G * (G + 1));
    const int D = G == 2 ? 4 : 6;
    if (wd < TR_PC * (D * D + 2)) wd = TR_PC * (D * D + 2);          // tangent_apply stages the tangent rows through the whole region (TangentRows)
    return (wd + 1) & ~1;
}

// transposed dofmap on the device, built once per mesh from the host copy of the dofmap
int ensure_transpose(dxo_ctx* ctx, dxo_mesh* m) {
    if (m->d_node_ptr) return DXO_OK;
    const int64_t nc = m->num_cells, nd = m->dev.ndofs, nn = m->num_field_nodes;
    if (nc * nd >= ((int64_t)1 << 32)) return DXO_E_SIZE;      // uint32 entries; the caller falls back to atomics
    std::vector<int64_t> ptr((size_t)nn + 1, 0);
    for (int64_t e = 0; e < nc * nd; ++e) ++ptr[(size_t)m->h_dofmap[(size_t)e] + 1];
    for (int64_t n = 0; n < nn; ++n) ptr[(size_t)n + 1] += ptr[(size_t)n];
    std::vector<uint32_t> ent((size_t)(nc * nd));
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t e = 0; e < nc * nd; ++e)      // visited in ascending (cell, a): a fixed order per node; stored as the fe index a*nc + cell
        ent[(size_t)fill[(size_t)m->h_dofmap[(size_t)e]]++] = (uint32_t)((e % nd) * nc + e / nd);
    DXO_HIP(ctx, hipMalloc((void**)&m->d_node_ptr, ptr.size() * sizeof(int64_t)));
    DXO_HIP(ctx, hipMalloc((void**)&m->d_node_ent, (ent.size() + 4) * sizeof(uint32_t)));   // + 4: node_sum reads four indices at a time
    DXO_HIP(ctx, hipMemcpy(m->d_node_ptr, ptr.data(), ptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    if (!ent.empty()) DXO_HIP(ctx, hipMemcpy(m->d_node_ent, ent.data(), ent.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return DXO_OK;
}

#ifdef DXO_EXPERIMENTS
#define DXO_PATCH_PART 2
#include "../../scripts/exp/adjoint_patch_kernels.h"
#undef DXO_PATCH_PART
#endif

// element-vector buffer for `bs` components, or nullptr when the call has to use atomics
double* two_pass_buffer(dxo_ctx* ctx, dxo_mesh* m, int bs, const int32_t* cells, int64_t n_cells) {
    if (ctx->adjoint_atomics || cells || n_cells != m->num_cells) return nullptr;
    if (ensure_transpose(ctx, m) != DXO_OK) return nullptr;
    const size_t need = (size_t)m->num_cells * m->dev.ndofs * bs * sizeof(double);
    if (m->fe_cap < need) {
        if (m->d_fe) (void)hipFree(m->d_fe);
        m->d_fe = nullptr;
        m->fe_cap = 0;
        if (hipMalloc((void**)&m->d_fe, need) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        m->fe_cap = need;
    }
    return m->d_fe;
}

// option consumer_overwrite with the atomics form of the scatter (entity subsets, adjoint_atomics): the vector is cleared first
int clear_for_atomics(dxo_ctx* ctx, const dxo_mesh* m, int bs, double* out, const double* fe, hipStream_t s) {
    if (fe || !ctx->consumer_overwrite) return DXO_OK;
    DXO_HIP(ctx, hipMemsetAsync(out, 0, (size_t)m->num_field_nodes * bs * sizeof(double), s));
    return DXO_OK;
}

void launch_node_sum(const dxo_ctx* ctx, const dxo_mesh* m, int bs, double* out, hipStream_t s) {
    int64_t blocks = (m->num_field_nodes + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * DXO_NS_BLOCKS_PER_CU;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (bs == 1) hipLaunchKernelGGL(node_sum<1>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, m->num_field_nodes, m->d_node_ptr, m->d_node_ent, m->d_fe, out, (int)(ctx->consumer_overwrite != 0));
    else if (bs == 2) hipLaunchKernelGGL(node_sum<2>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, m->num_field_nodes, m->d_node_ptr, m->d_node_ent, m->d_fe, out, (int)(ctx->consumer_overwrite != 0));
    else hipLaunchKernelGGL(node_sum<3>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, m->num_field_nodes, m->d_node_ptr, m->d_node_ent, m->d_fe, out, (int)(ctx->consumer_overwrite != 0));
}

template <int G, int BS, int KIND>
void launch_adjoint(const dxo_ctx* ctx, const dxo_mesh* m, const double* S, const int32_t* cells, int64_t n_cells,
                    double* out, double* fe, hipStream_t s) {
    const int wd = adjoint_lds_wave(m);
    const int64_t n_groups = (n_cells + m->dev.cells_per_wave - 1) / m->dev.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;
    const size_t shm = (size_t)(m->dev.table_doubles + 4 * wd) * sizeof(double);
    hipLaunchKernelGGL((operand_adjoint<G, BS, KIND>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, m->dev, m->d_wq, wd, S,
                       cells, n_cells, out, fe);
}

template <int G, int BS>
int dispatch_adjoint(const dxo_ctx* ctx, const dxo_mesh* m, int kind, const double* S, const int32_t* cells,
                     int64_t n_cells, double* out, double* fe, hipStream_t s) {
    switch (kind) {
        case DXO_OPERAND_VALUE: launch_adjoint<G, BS, DXO_OPERAND_VALUE>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK;
        case DXO_OPERAND_GRAD: launch_adjoint<G, BS, DXO_OPERAND_GRAD>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK;
        case DXO_OPERAND_VALUE_GRAD: launch_adjoint<G, BS, DXO_OPERAND_VALUE_GRAD>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK;
        case DXO_OPERAND_EPS_MANDEL:
            if constexpr (BS == G) { launch_adjoint<G, BS, DXO_OPERAND_EPS_MANDEL>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DEFGRAD:
            if constexpr (BS == G) { launch_adjoint<G, BS, DXO_OPERAND_DEFGRAD>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DIV:
            if constexpr (BS == G) { launch_adjoint<G, BS, DXO_OPERAND_DIV>(ctx, m, S, cells, n_cells, out, fe, s); return DXO_OK; }
            return DXO_E_DIM;
    }
    return DXO_E_OPTION;
}

}  // namespace

extern "C" int dxo_mesh_set_weights(dxo_ctx* ctx, dxo_mesh* mesh, const double* weights) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!mesh || !weights) return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_set_weights: NULL argument");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (!mesh->d_wq) DXO_HIP(ctx, hipMalloc((void**)&mesh->d_wq, (size_t)mesh->dev.nq * sizeof(double)));
    DXO_HIP(ctx, hipMemcpy(mesh->d_wq, weights, (size_t)mesh->dev.nq * sizeof(double), hipMemcpyHostToDevice));
    return DXO_OK;
}

// is ctx option adjoint_patch = 1 available in this build? (dxo_ctx_set_option asks; the patch form lives in scripts/exp/)
bool dxo_adjoint_patch_available() {
#ifdef DXO_EXPERIMENTS
    return true;
#else
    return false;
#endif
}

#ifdef DXO_EXPERIMENTS
#define DXO_PATCH_PART 3
#include "../../scripts/exp/adjoint_patch_kernels.h"
#undef DXO_PATCH_PART
#endif

extern "C" int dxo_operand_adjoint(dxo_ctx* ctx, dxo_mesh* mesh, int kind, int bs, const double* S, const int32_t* cells,
                                   int64_t n_cells, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_operand_adjoint: mesh is NULL");
    if (!mesh->d_wq) return dxo_fail(ctx, DXO_E_OPTION, "dxo_operand_adjoint: quadrature weights not set (dxo_mesh_set_weights)");
    if (kind == DXO_OPERAND_CAUCHY_GREEN || kind == DXO_OPERAND_I1 || kind == DXO_OPERAND_DETF)
        return dxo_fail(ctx, DXO_E_OPTION, "dxo_operand_adjoint: a nonlinear operand (C, I1, det F) has no adjoint — its linearisation is a form UFL derives on the reference side");
    const int D = dxo_operand_value_size(mesh->gdim, bs, kind);
    if (D == DXO_E_OPTION) return dxo_fail(ctx, DXO_E_OPTION, "dxo_operand_adjoint: unknown operand kind");
    if (D < 0 || (bs != 1 && bs != mesh->gdim)) return dxo_fail(ctx, DXO_E_DIM, "dxo_operand_adjoint: block size does not fit the operand kind / gdim (the adjoint takes bs = 1 or gdim)");
    if (!cells) n_cells = n_cells < 0 ? mesh->num_cells : n_cells;
    if (n_cells < 0 || (!cells && n_cells > mesh->num_cells)) return dxo_fail(ctx, DXO_E_SIZE, "dxo_operand_adjoint: bad n_cells");
    if (n_cells == 0) return DXO_OK;
    if (!S || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_operand_adjoint: NULL array");
    if ((size_t)(mesh->dev.table_doubles + 4 * adjoint_lds_wave(mesh)) * sizeof(double) > 64 * 1024)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_operand_adjoint: element too large for the LDS budget");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const bool c8 = kind == DXO_OPERAND_EPS_MANDEL && ctx->adjoint_cell && !cells && n_cells == mesh->num_cells && mesh->gdim == 3 && bs == 3 &&
                    mesh->dev.nq == 8 && mesh->dev.ngeom == 8 && (mesh->dev.ndofs == 27 || mesh->dev.ndofs == 8) && (((uintptr_t)S & 15u) == 0);
#ifdef DXO_EXPERIMENTS
    if (c8 && use_patches(ctx, mesh, bs, cells, n_cells)) {
        // hexahedra with the 2x2x2 rule, patch form: the entries meet in LDS, only patch-boundary partials travel (adjoint_patch.h)
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        const size_t shm = (size_t)(C8_LDS + PATCH_WAVES * mesh->patch.dev.max_priv * 3) * sizeof(double);
        const int blocks = patch_grid(ctx, mesh, 6);
        const int ow = (int)(ctx->consumer_overwrite != 0);
        if (mesh->dev.ndofs == 27) hipLaunchKernelGGL((operand_adjoint_c8_patch<27>), dim3(blocks), dim3(PATCH_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, mesh->patch.dev, out, ow);
        else                       hipLaunchKernelGGL((operand_adjoint_c8_patch<8>), dim3(blocks), dim3(PATCH_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, mesh->patch.dev, out, ow);
        launch_node_sum_patch(ctx, mesh, bs, out, s);
        return dxo_device_end(ctx, s);
    }
#endif
    double* fe = two_pass_buffer(ctx, mesh, bs, cells, n_cells);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    rc = clear_for_atomics(ctx, mesh, bs, out, fe, s);
    if (rc != DXO_OK) return rc;
    if (kind == DXO_OPERAND_EPS_MANDEL && ctx->adjoint_cell && !cells && n_cells == mesh->num_cells && mesh->gdim == 3 && bs == 3 &&
        mesh->dev.nq == 8 && mesh->dev.ngeom == 8 && (mesh->dev.ndofs == 27 || mesh->dev.ndofs == 8) && (((uintptr_t)S & 15u) == 0)) {
        // hexahedra with the 2x2x2 rule: contraction across the cell's lanes, nothing staged in LDS (operand_adjoint_c8)
        const int64_t n_groups = (n_cells + 7) / 8;
        int64_t blocks = (n_groups + 3) / 4;
        const int64_t cap = (int64_t)ctx->compute_units * DXO_C8_ADJ_BLOCKS_PER_CU;
        if (blocks > cap) blocks = cap;
        blocks = (blocks + 7) / 8 * 8;
        const size_t shm = (size_t)(C8_LDS + (ctx->adjoint_mfma ? (DXO_BLOCK / DXO_WAVE) * C8M_WAVE : 0)) * sizeof(double);
        if (ctx->adjoint_mfma) {      // the contraction on the f64 matrix pipe
            if (mesh->dev.ndofs == 27) hipLaunchKernelGGL((operand_adjoint_c8_mfma<27>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, n_cells, out, fe);
            else                       hipLaunchKernelGGL((operand_adjoint_c8_mfma<8>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, n_cells, out, fe);
        } else if (mesh->dev.ndofs == 27) hipLaunchKernelGGL((operand_adjoint_c8<27>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, n_cells, out, fe);
        else                       hipLaunchKernelGGL((operand_adjoint_c8<8>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, S, n_cells, out, fe);
        if (fe) launch_node_sum(ctx, mesh, bs, out, s);
        return dxo_device_end(ctx, s);
    }
    if (fe && kind == DXO_OPERAND_EPS_MANDEL && ctx->adjoint_cell && launch_adjoint_cell_eps(ctx, mesh, S, fe, s)) {
        launch_node_sum(ctx, mesh, bs, out, s);      // lane = cell form (adjoint_cell.h) for the standard elements
        return dxo_device_end(ctx, s);
    }
    if (mesh->gdim == 2) rc = bs == 1 ? dispatch_adjoint<2, 1>(ctx, mesh, kind, S, cells, n_cells, out, fe, s) : dispatch_adjoint<2, 2>(ctx, mesh, kind, S, cells, n_cells, out, fe, s);
    else                 rc = bs == 1 ? dispatch_adjoint<3, 1>(ctx, mesh, kind, S, cells, n_cells, out, fe, s) : dispatch_adjoint<3, 3>(ctx, mesh, kind, S, cells, n_cells, out, fe, s);
    if (rc != DXO_OK) return dxo_fail(ctx, rc, "dxo_operand_adjoint: unsupported (gdim, bs, kind)");
    if (fe) launch_node_sum(ctx, mesh, bs, out, s);
    return dxo_device_end(ctx, s);
}

namespace {

// shared body of dxo_tangent_diagonal / dxo_tangent_diagonal_vm (vs == nullptr: rows of C_tang)
int tangent_diagonal_impl(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, const VmStateSrc* vs, double* out, const char* who) {
    if (!mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_tangent_diagonal: mesh is NULL");
    if (!mesh->d_wq) return dxo_fail(ctx, DXO_E_OPTION, "dxo_tangent_diagonal: quadrature weights not set (dxo_mesh_set_weights)");
    if (mesh->num_cells == 0) return DXO_OK;
    if ((!vs && !C_tang) || (vs && (!vs->sigma || !vs->dp)) || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_tangent_diagonal: NULL array");
    if (((uintptr_t)(vs ? (const void*)vs->sigma : (const void*)C_tang) & 15u) != 0)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_tangent_diagonal: C_tang / sigma must be 16-byte aligned");
    (void)who;
    const bool rs = DXO_TA_RS && mesh->gdim == 3 && mesh->dev.ndofs == 27 && mesh->dev.ngeom == 8 && mesh->dev.nq == 8;   // Q2 hexahedra, 2x2x2 rule
    // MFMA form: the wave's slice is the vertex buffer + the staged matrices of c8m_contract (state-based), or the staging space of the
    // tangent rows, which also holds them (C_tang rows: 66 KB per workgroup with the product tables — above the 64 KB a launch gets without asking)
    // scatter_mfma.h: P2 tetrahedra (4-point rule) and P2 triangles (3-point rule), state-based form
    const bool gm_tet = vs && ctx->adjoint_mfma && mesh->gdim == 3 && mesh->dev.ndofs == 10 && mesh->dev.ngeom == 4 && mesh->dev.nq == 4;
    const bool gm_tri = vs && ctx->adjoint_mfma && mesh->gdim == 2 && mesh->dev.ndofs == 6 && mesh->dev.ngeom == 3 && mesh->dev.nq == 3;
    const bool q1 = vs && ctx->adjoint_mfma && DXO_TA_RS && mesh->gdim == 3 && mesh->dev.ndofs == 8 && mesh->dev.ngeom == 8 && mesh->dev.nq == 8;   // Q1 hexahedra, state-based
    bool mf = (rs || q1) && ctx->adjoint_mfma;
    if (mf && !vs) {
        // the C_tang-rows form needs the raised launch limit of its one instantiation; a runtime that refuses it gets the DPP form
        static std::atomic<uint64_t> raised{0}, refused{0};           // one bit per device: the attribute belongs to the device's copy of the kernel
        const uint64_t bit = 1ull << (ctx->device & 63);
        if (!((raised.load() | refused.load()) & bit)) {
            DXO_HIP(ctx, hipSetDevice(ctx->device));
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tangent_diag<3, 27, false, true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e == hipSuccess) raised.fetch_or(bit);
            else { (void)hipGetLastError(); refused.fetch_or(bit); }
        }
        if (refused.load() & bit) mf = false;
    }
    int wd = (mf && vs) ? ((mesh->dev.cells_per_wave * op_odd(mesh->dev.ngeom * 3) + 1) & ~1) + C8M_WAVE : diag_lds_wave(mesh);
    const int gm_stage = gm_tet ? GmShape<3, 10, 4, 3>::STAGE : gm_tri ? GmShape<2, 6, 3, 3>::STAGE : 0;
    const int gm_tab = gm_tet ? 2 * GmShape<3, 10, 4, 3>::ATAB : gm_tri ? GmShape<2, 6, 3, 3>::ATAB : 0;
    if (gm_stage) {      // the staging slice behind the vertex buffer
        const int need = ((mesh->dev.cells_per_wave * op_odd(mesh->dev.ngeom * mesh->gdim) + 1) & ~1) + gm_stage;
        if (wd < need) wd = (need + 1) & ~1;
    }
    const size_t shm = (size_t)(mesh->dev.table_doubles + 4 * wd + ((rs || q1) ? (mf ? 2 * 12 * DXO_WAVE : C8_LDS) : gm_tab)) * sizeof(double);
    if (shm > (mf && !vs ? 80 : 64) * 1024) return dxo_fail(ctx, DXO_E_SIZE, "dxo_tangent_diagonal: element too large for the LDS budget");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    double* fe = two_pass_buffer(ctx, mesh, mesh->gdim, nullptr, mesh->num_cells);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    rc = clear_for_atomics(ctx, mesh, mesh->gdim, out, fe, s);
    if (rc != DXO_OK) return rc;
    const int64_t n_groups = (mesh->num_cells + mesh->dev.cells_per_wave - 1) / mesh->dev.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * (vs ? DXO_TD_VM_BLOCKS_PER_CU : DXO_TD_BLOCKS_PER_CU);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;
    const VmStateSrc none{};
    const VmStateSrc& src = vs ? *vs : none;
#define DXO_DIAG_LAUNCH(...) hipLaunchKernelGGL((tangent_diag<__VA_ARGS__>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, wd, C_tang, src, mesh->num_cells, out, fe)
    if (gm_tri)          { DXO_DIAG_LAUNCH(2, 6, true, true); }
    else if (gm_tet)     { DXO_DIAG_LAUNCH(3, 10, true, true); }
    else if (mesh->gdim == 2) { if (vs) DXO_DIAG_LAUNCH(2, 0, true); else DXO_DIAG_LAUNCH(2, 0, false); }
    else if (q1)         { DXO_DIAG_LAUNCH(3, 8, true, true); }
    else if (mf)         { if (vs) DXO_DIAG_LAUNCH(3, 27, true, true); else DXO_DIAG_LAUNCH(3, 27, false, true); }
    else if (rs)         { if (vs) DXO_DIAG_LAUNCH(3, 27, true); else DXO_DIAG_LAUNCH(3, 27, false); }
    else                 { if (vs) DXO_DIAG_LAUNCH(3, 0, true); else DXO_DIAG_LAUNCH(3, 0, false); }
#undef DXO_DIAG_LAUNCH
    if (fe) launch_node_sum(ctx, mesh, mesh->gdim, out, s);
    return dxo_device_end(ctx, s);
}

// shared body of dxo_tangent_apply / dxo_tangent_apply_vm
int tangent_apply_impl(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, const VmStateSrc* vs, const double* v, double* out) {
    if (!mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_tangent_apply: mesh is NULL");
    if (!mesh->d_wq) return dxo_fail(ctx, DXO_E_OPTION, "dxo_tangent_apply: quadrature weights not set (dxo_mesh_set_weights)");
    if (mesh->num_cells == 0) return DXO_OK;
    if ((!vs && !C_tang) || (vs && (!vs->sigma || !vs->dp)) || !v || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_tangent_apply: NULL array");
    if (((uintptr_t)(vs ? (const void*)vs->sigma : (const void*)C_tang) & 15u) != 0)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_tangent_apply: C_tang / sigma must be 16-byte aligned");
    const bool rs = DXO_TA_RS && mesh->gdim == 3 && mesh->dev.ndofs == 27 && mesh->dev.ngeom == 8 && mesh->dev.nq == 8;   // Q2 hexahedra, 2x2x2 rule
    const bool c8 = rs && DXO_TA_C8_FORWARD && !vs;
    // Q1 hexahedra with the 2x2x2 rule, state-based form: the same kernel with 8 nodes (matrix-pipe scatter only)
    const bool q1 = vs && ctx->adjoint_mfma && DXO_TA_RS && mesh->gdim == 3 && mesh->dev.ndofs == 8 && mesh->dev.ngeom == 8 && mesh->dev.nq == 8;
    // the state form stages nothing: its wave region is the gather buffer (and the parked tensors where the scatter uses them)
    int wd = (rs || q1) ? (vs ? ((mesh->dev.cells_per_wave * (op_odd(mesh->dev.ndofs * 3) + op_odd(mesh->dev.ngeom * 3)) + 1) & ~1) : apply_rs_lds_wave(mesh))
                        : adjoint_lds_wave(mesh);
    if (q1 && wd < C8M_WAVE) wd = C8M_WAVE;      // c8m_contract stages T in the wave's region
    // scatter_mfma.h: P2 tetrahedra (4-point rule) and P2 triangles (3-point rule), state-based form (with the tangent rows' registers on top the
    // compile-time element costs more than the scatter gains: 0.511 against 0.503 ms on triangles, 1.135 / 1.106 on tetrahedra)
    const bool gm_tet = vs && ctx->adjoint_mfma && mesh->gdim == 3 && mesh->dev.ndofs == 10 && mesh->dev.ngeom == 4 && mesh->dev.nq == 4;
    const bool gm_tri = vs && ctx->adjoint_mfma && mesh->gdim == 2 && mesh->dev.ndofs == 6 && mesh->dev.ngeom == 3 && mesh->dev.nq == 3;
    const int gm_tab = gm_tet ? GmShape<3, 10, 4>::ATAB : gm_tri ? GmShape<2, 6, 3>::ATAB : 0;
    const size_t shm = c8 ? (size_t)(C8_LDS + 4 * TangentRows<6>::LDS_DOUBLES) * sizeof(double)
                          : (size_t)(mesh->dev.table_doubles + 4 * wd + ((rs || q1) ? (ctx->adjoint_mfma ? 12 * DXO_WAVE + C8M_FTAB : C8_LDS) : gm_tab)) * sizeof(double);
    if (shm > 64 * 1024) return dxo_fail(ctx, DXO_E_SIZE, "dxo_tangent_apply: element too large for the LDS budget");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    double* fe = two_pass_buffer(ctx, mesh, mesh->gdim, nullptr, mesh->num_cells);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    rc = clear_for_atomics(ctx, mesh, mesh->gdim, out, fe, s);
    if (rc != DXO_OK) return rc;
    const int64_t n_groups = (mesh->num_cells + mesh->dev.cells_per_wave - 1) / mesh->dev.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * (vs ? DXO_TA_VM_BLOCKS_PER_CU : DXO_TA_BLOCKS_PER_CU);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;
    const VmStateSrc none{};
    const VmStateSrc& src = vs ? *vs : none;
#define DXO_APPLY_LAUNCH(...) hipLaunchKernelGGL((tangent_apply<__VA_ARGS__>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, wd, C_tang, src, v, mesh->num_cells, out, fe)
#ifdef DXO_EXPERIMENTS
    if (!vs && DXO_TANGENT_CELL && fe && ctx->adjoint_cell && launch_tangent_cell(ctx, mesh, C_tang, v, fe, s)) {
        // lane = cell form for P2 triangles — correct, but 0.61 against 0.55-0.58 ms per 10^7 points for the wave-group kernel
    } else if (c8)
        hipLaunchKernelGGL((tangent_apply_c8<27>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, mesh->dev, mesh->d_wq, C_tang, v, mesh->num_cells, out, fe);
    else
#endif
    if (gm_tri)               { DXO_APPLY_LAUNCH(2, 6, 3, true, true); }
    else if (gm_tet)          { DXO_APPLY_LAUNCH(3, 10, 4, true, true); }
    else if (mesh->gdim == 2) { if (vs) DXO_APPLY_LAUNCH(2, 0, 0, true); else DXO_APPLY_LAUNCH(2, 0, 0, false); }
    else if (q1)              { DXO_APPLY_LAUNCH(3, 8, 8, true, true); }
    else if (rs && ctx->adjoint_mfma) { if (vs) DXO_APPLY_LAUNCH(3, 27, 8, true, true); else DXO_APPLY_LAUNCH(3, 27, 8, false, true); }
    else if (rs)              { if (vs) DXO_APPLY_LAUNCH(3, 27, 8, true); else DXO_APPLY_LAUNCH(3, 27, 8, false); }   // Q2 hexahedra, 2x2x2 rule: compile-time trip counts, scatter in registers
    else                      { if (vs) DXO_APPLY_LAUNCH(3, 0, 0, true); else DXO_APPLY_LAUNCH(3, 0, 0, false); }
#undef DXO_APPLY_LAUNCH
    if (fe) launch_node_sum(ctx, mesh, mesh->gdim, out, s);
    return dxo_device_end(ctx, s);
}

bool vm_state_src(dxo_ctx* ctx, const dxo_mesh* mesh, const dxo_vm_params* prm, const double* sigma, const double* dp, VmStateSrc& out) {
    if (!mesh || !prm) { dxo_fail(ctx, DXO_E_NULL, "dxo_tangent_*_vm: NULL mesh or params"); return false; }
    out.c = make_const(*prm);
    out.sigma = sigma;
    out.dp = dp;
    return true;
}

}  // namespace

// Residual of a von Mises Newton iteration in ONE call: (sigma, dp) = return map(eps(u), sigma_n, p) and R += sum_q w|J| B^T sigma.
// dxo_von_mises_field (no tangent) and dxo_operand_adjoint back to back; with option vm_residual_fused = 1, on Q2 hexahedra with the
// 2x2x2 rule, one kernel (vm_field RES) scatters the stress while it is in registers — an experiment that measured no faster.
extern "C" int dxo_von_mises_residual(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, const double* u, const double* sigma_n,
                                      const double* p, double* sigma, double* dp, double* R) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_residual: NULL params or mesh");
    if (!mesh->d_wq) return dxo_fail(ctx, DXO_E_OPTION, "dxo_von_mises_residual: quadrature weights not set (dxo_mesh_set_weights)");
    if (mesh->num_cells == 0) return DXO_OK;
    if (!u || !sigma_n || !p || !sigma || !dp || !R) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_residual: NULL array");
    if (((uintptr_t)u | (uintptr_t)p | (uintptr_t)dp | (uintptr_t)R) & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_residual: arrays must be 8-byte aligned");
    if (((uintptr_t)sigma_n | (uintptr_t)sigma) & 15u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_residual: sigma_n, sigma must be 16-byte aligned");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    double* fe = (ctx->vm_residual_fused && dxo_vmf_residual_eligible(mesh)) ? two_pass_buffer(ctx, mesh, 3, nullptr, mesh->num_cells) : nullptr;
    if (!fe) {
        int rc = dxo_von_mises_field(ctx, prm, mesh, DXO_MEM_DEVICE, u, sigma_n, p, nullptr, sigma, dp);
        if (rc != DXO_OK) return rc;
        return dxo_operand_adjoint(ctx, mesh, DXO_OPERAND_EPS_MANDEL, mesh->gdim, sigma, nullptr, -1, R);
    }
    hipStream_t s = dxo_launch_stream(ctx);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    rc = dxo_vmf_residual_launch(ctx, prm, mesh, u, sigma_n, p, sigma, dp, fe, s);
    if (rc != DXO_OK) return rc;
    launch_node_sum(ctx, mesh, 3, R, s);
    return dxo_device_end(ctx, s);
}

extern "C" int dxo_tangent_diagonal(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    return tangent_diagonal_impl(ctx, mesh, C_tang, nullptr, out, "dxo_tangent_diagonal");
}

extern "C" int dxo_tangent_apply(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, const double* v, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    return tangent_apply_impl(ctx, mesh, C_tang, nullptr, v, out);
}

extern "C" int dxo_tangent_apply_vm(dxo_ctx* ctx, dxo_mesh* mesh, const dxo_vm_params* prm, const double* sigma, const double* dp,
                                    const double* v, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    VmStateSrc vs;
    if (!vm_state_src(ctx, mesh, prm, sigma, dp, vs)) return DXO_E_NULL;
    return tangent_apply_impl(ctx, mesh, nullptr, &vs, v, out);
}

extern "C" int dxo_tangent_diagonal_vm(dxo_ctx* ctx, dxo_mesh* mesh, const dxo_vm_params* prm, const double* sigma, const double* dp,
                                       double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    VmStateSrc vs;
    if (!vm_state_src(ctx, mesh, prm, sigma, dp, vs)) return DXO_E_NULL;
    return tangent_diagonal_impl(ctx, mesh, nullptr, &vs, out, "dxo_tangent_diagonal_vm");
}
