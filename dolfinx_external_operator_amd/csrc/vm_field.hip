// vm_field.hip — operand evaluation fused in front of the von Mises return map.
//
// The reference's residual evaluation does, per SNES iteration (demo_plasticity_von_mises.py:445-456):
//   evaluated_operands = evaluate_operands(F_external_operators)          # deps = eps(Du) at every quadrature point
//   ((_, sigma_new, dp_new),) = evaluate_external_operators(J_external_operators, evaluated_operands)
// i.e. Expression.eval writes deps (num_cells, nq, d) and the constitutive kernel reads it back. Here one launch
// does both: the wave gathers its cells' displacement dofs, each lane forms the strain increment of its own
// quadrature point in registers (operand_core.h) and runs the radial return on it (vm_core.h); deps never exists
// in memory. Traffic per point at d = 6 on Q2 hexahedra: read sigma_n 48 + p 8 + ~45 B of dofs/dofmaps,
// write 344 B — against 93 B + 448 B for the two separate launches.
//
// Tiling: a wave owns floor(64 / nq) consecutive cells = up to 64 consecutive points (63 for nq = 3); sigma_n,
// sigma and C_tang move exactly as in vm_tile (lane-linear 16-byte accesses through wave-private LDS, tangent
// rebuilt in output order). The operand's gather buffer and vm_tile's staging slices share the wave's LDS region.
#include "dxo_common.h"
#include "operand_core.h"
#include "vm_core.h"
#include "vm_host.h"
#include "cell8_dpp.h"
#include "cell8_mfma.h"

namespace {

#ifndef DXO_VMF_WAVES
#define DXO_VMF_WAVES 3
#endif
#ifndef DXO_OP_CT
#define DXO_OP_CT 1
#endif
#ifndef DXO_VMF_PRELOAD
#define DXO_VMF_PRELOAD 0   // request sigma_n / p before the strain is formed: no gain (0.844 vs 0.844 ms with two waves per SIMD, 1.00 with three: spills)
#endif
#ifndef DXO_VMF_DMA
#define DXO_VMF_DMA 1     // sigma_n of the group lands in a slice of its own by global_load_lds (lane-linear, no registers) while the strain is formed:
                          // Q2 hexahedra, (sigma, dp)-only launch 0.467 -> 0.414 ms, with tangent 0.935 -> 0.924 (one lease, scripts/exp/vmfield_ab.py); 0 = the
                          // loads are issued after the contraction and awaited at once (in registers ahead of it they cost the third wave: DXO_VMF_PRELOAD)
#endif
#ifndef DXO_VMF_ROWS
#define DXO_VMF_ROWS 0   // 1: tangent rows built per lane and turned to output order through LDS (vm_store_tangent_rows: ~90 instead of ~650 vector
                         // instructions per tile, bit-identical output) — 0.836 against 0.832 ms: the walk's arithmetic is not what the kernel waits for
#endif
#ifndef DXO_VMF_RES_MFMA
#define DXO_VMF_RES_MFMA 1   // residual form: the scatter on the matrix pipe (cell8_mfma.h) instead of the DPP reduce-scatter
#endif
#ifndef DXO_VMF_RES_WAVES
#define DXO_VMF_RES_WAVES 2   // residual form: 224 registers. Three waves per SIMD spill 136 registers even with w|J|J^-1 parked in LDS (1.80 ms
                              // against 1.19 for the whole call): the scatter's 8 table rows and partials sit on top of the gather pipeline
#endif
#ifndef DXO_VMF_STATE2D_WAVES
#define DXO_VMF_STATE2D_WAVES 4   // (sigma, dp)-only launch in 2-D: 132 registers at three waves per SIMD
#endif
#ifndef DXO_VMF_STATE3D_WAVES
#define DXO_VMF_STATE3D_WAVES 3   // (sigma, dp)-only launch in 3-D: waves per SIMD its register budget is set for (see profiles/r05_vm_field_state.txt)
#endif
#ifndef DXO_VMF_FULL
#define DXO_VMF_FULL 1   // guard-free tangent stores for full groups: -0.5 % (0.812 vs 0.816 ms); grid of 8 / 16 / 32 / 64 workgroups per CU: 0.819 / 0.816 / 0.827 / 0.867
#endif
#ifndef DXO_VMF_STATE_BLOCKS_PER_CU
#define DXO_VMF_STATE_BLOCKS_PER_CU 8     // grid of the (sigma, dp)-only launch (arithmetic-bound, unlike the full one): 8 against 16 / 32 / 64 workgroups per CU:
                                          // hexahedra 0.456 / 0.465 / 0.483 / 0.501 ms, triangles 0.239 / 0.254 / 0.262 / 0.296; 3 - 12 within the noise of each other
#endif
#ifndef DXO_VMF_BLOCKS_PER_CU
#define DXO_VMF_BLOCKS_PER_CU 16
#endif
// RES (hexahedra with the 2x2x2 rule, cell8_mfma.h; option vm_residual_fused, off by default): the kernel also forms the internal force of the stress it has just returned —
// element vectors fe[node][cell][i] = sum_q w|J| B^T sigma, reduce-scattered over the cell's 8 lanes while sigma, J^-1 and |J| are
// still in registers (dxo_von_mises_residual; node_sum follows). No tangent is written in this form.
// MODE 0: (C_tang, sigma, dp). MODE 1: (sigma, dp) only — its own instantiation, so that the profiler's kernel names tell the two
// launches apart (the arithmetic of the stores that remain is the same). MODE 2: RES.
// does the instantiation (MODE, ND_CT) fetch the old state by global_load_lds when its slices fit? (host and device agree through this one function)
constexpr bool vmf_dma_compiled(int mode, int nd_ct) { return DXO_VMF_DMA && !DXO_VMF_PRELOAD && !(mode == 0 && nd_ct == 27); }

template <int G, bool NT, int ND_CT = 0, int NG_CT = 0, int MODE = 0, bool DMA_OK = true>
__global__ __launch_bounds__(DXO_BLOCK, MODE == 2 ? DXO_VMF_RES_WAVES : (MODE == 1 && G == 2) ? DXO_VMF_STATE2D_WAVES : (MODE == 1) ? DXO_VMF_STATE3D_WAVES : DXO_VMF_WAVES) void vm_field(VmConst c, OperandDev m, int wave_doubles, int64_t cell0,
                                                         int64_t n_cells, const double* __restrict__ u,
                                                         const double* __restrict__ sigma_n,
                                                         const double* __restrict__ p, double* __restrict__ C_tang,
                                                         double* __restrict__ sigma, double* __restrict__ dp_out,
                                                         const double* __restrict__ wq, double* __restrict__ fe) {
    constexpr bool RES = MODE == 2;
    // the stress tile by global_load_lds (DXO_VMF_DMA) — not in the one instantiation that sits exactly at its register budget (Q2 hexahedra with
    // tangent: 168 registers; the two address pairs the copies need spill there and cost 2 %), and not when the host found that the landing slices
    // do not fit the workgroup's 64 KB beside the element's tables and gather regions (DMA_OK = false: e.g. Q2 hexahedra with a 27-point rule)
    constexpr bool DMA = DMA_OK && vmf_dma_compiled(MODE, ND_CT);
    static_assert(!RES || (G == 3 && ND_CT > 0 && ND_CT <= C8_NODES && NG_CT == 8), "the residual form is the eight-point hexahedron's");
    constexpr int D = G == 2 ? 4 : 6;
    using T = VmTile<D>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    double* tab8 = lds + m.table_doubles + T::WAVES * wave_doubles;     // RES: the padded dphi rows c8_scatter reads
    if constexpr (RES) {
        if constexpr (DXO_VMF_RES_MFMA) c8m_fill_A<ND_CT>(m, tab8);
        else c8_fill_tables(m, tab8);
    }
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * wave_doubles;   // operand gather buffer, then X | Y of vm_tile
    double* X = W;
    double* Y = X + T::X_DOUBLES;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
    dxo_f64x2* N2 = RES ? X2 : Y2;      // where sigma_n turns from lane-linear to point-per-lane (RES keeps w|J|J^-1 in Y)
    // landing slice of the wave (X_DOUBLES + 64 doubles behind every wave's region and the RES tables)
    dxo_f64x2* L2 = reinterpret_cast<dxo_f64x2*>(lds + m.table_doubles + T::WAVES * wave_doubles + (RES ? C8_LDS : 0) + wave * (T::X_DOUBLES + DXO_WAVE));
    uint32_t* P32 = reinterpret_cast<uint32_t*>(L2 + T::X_DOUBLES / 2);      // 64 doubles behind the stress slice
    const double w_l = RES ? wq[lane & 7] : 0.0;
    const int cpw = m.cells_per_wave;
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, T::WAVES, wave);
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    // the dof gather runs as a register pipeline two groups ahead (operand_core.h) whenever the element fits it
    const bool piped = operand_can_pipe(m);
    OperandPipe<G, G> pf;
    int64_t grp = walk.first;
    if (piped) {
        pipe_load_indices<G, G>(m, pf, cell0 + grp * cpw, cells_in(grp), lane);
        pipe_load_values<G, G>(m, pf, u);
        pipe_load_indices<G, G>(m, pf, cell0 + (grp + stride) * cpw, cells_in(grp + stride), lane);
    }
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;                 // first cell of the group, relative to cell0
        const int ncell = cells_in(grp);
        const int npts = ncell * m.nq;
        const int64_t p0 = c0 * m.nq;                 // first point, relative to the state/output arrays passed in
        const int nvec = npts * T::CH_VEC;

#if DXO_VMF_PRELOAD
        // the state of this group is requested BEFORE the strain is formed: its HBM latency runs under the contraction
        dxo_f64x2 sreg[T::CH_VEC];
        {
            const dxo_f64x2* g_s0 = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                const int idx = k * DXO_WAVE + lane;
                sreg[k] = idx < nvec ? g_s0[idx] : dxo_f64x2{0.0, 0.0};
            }
        }
        const double p_l = lane < npts ? p[p0 + lane] : 0.0;
#endif
        if constexpr (DMA) {
            // chunk q of the tile (16 bytes) goes to slot q of the slice: the wave's uniform base + lane x 16, as the instruction writes it. Lanes
            // beyond the tile's chunks are masked and leave their slots as they were: only idle lanes read those, and their results are never stored
            const dxo_f64x2* g_s0 = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k)
                if (k * DXO_WAVE + lane < nvec)
                    __builtin_amdgcn_global_load_lds(g_s0 + k * DXO_WAVE + lane, (__attribute__((address_space(3))) void*)(L2 + k * DXO_WAVE), 16, 0, 0);
            // p the same way, as its two 32-bit halves (the instruction moves 4, 12 or 16 bytes per lane): [low words of the 64 points | high words]
            if (lane < npts) {
                const uint32_t* g_p = reinterpret_cast<const uint32_t*>(p + p0 + lane);
                __builtin_amdgcn_global_load_lds(g_p, (__attribute__((address_space(3))) void*)P32, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(g_p + 1, (__attribute__((address_space(3))) void*)(P32 + DXO_WAVE), 4, 0, 0);
            }
        }
        // ---- A1: strain increment of this lane's point from the displacement dofs
        double e[D];
        bool active;
        double sK22 = 0.0;                  // RES only
        if (piped) {
            pipe_commit<G, G>(m, pf, W, ncell, lane);
            pipe_load_values<G, G>(m, pf, u);
            pipe_load_indices<G, G>(m, pf, cell0 + (grp + 2 * stride) * cpw, cells_in(grp + 2 * stride), lane);
            if constexpr (RES) {
                double Kinv[G][G], detJ;
                active = operand_compute_geo<G, G, DXO_OPERAND_EPS_MANDEL, ND_CT, NG_CT>(m, tab, W, ncell, lane, e, Kinv, detJ);
                // w |J| J^-1 waits for the returned stress in the wave's LDS slice (behind the 384 doubles the stress rows are staged in),
                // not in registers: the return map and the scatter each need the file at three waves per SIMD
                const double scale = w_l * fabs(detJ);
#pragma unroll
                for (int k = 0; k < 8; ++k) Y[k * DXO_WAVE + lane] = scale * Kinv[k / 3][k % 3];
                sK22 = scale * Kinv[2][2];
            } else {
                active = operand_compute<G, G, DXO_OPERAND_EPS_MANDEL, ND_CT, NG_CT>(m, tab, W, ncell, lane, e);
            }
        } else {
            active = operand_point<G, G, DXO_OPERAND_EPS_MANDEL>(m, tab, W, u, nullptr, cell0 + c0, ncell, lane, e);
        }
        if (!active) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[k] = 0.0;
        }
        // ---- A2: sigma_n lane-linear -> LDS -> point-per-lane
#if DXO_VMF_PRELOAD
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) N2[k * DXO_WAVE + lane] = sreg[k];
#else
        const dxo_f64x2* g_s = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
        double p_l = 0.0;
        if constexpr (!DMA) {
#pragma unroll
            for (int k = 0; k < T::CH_VEC; ++k) {
                const int idx = k * DXO_WAVE + lane;
                N2[idx] = idx < nvec ? g_s[idx] : dxo_f64x2{1.0 + lane, 0.5};
            }
            p_l = lane < npts ? p[p0 + lane] : 0.0;
        }
#endif
        const dxo_f64x2* S2 = DMA ? L2 : N2;
        if constexpr (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the tile has landed (the strain's arithmetic ran meanwhile)
            if (lane < npts) {
                const uint32_t lo = P32[lane], hi = P32[DXO_WAVE + lane];
                p_l = __hiloint2double((int)hi, (int)lo);
            }
        }
        wave_lds_fence();
        double sn[D];
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const dxo_f64x2 b2 = S2[lane * T::CH_VEC + k];
            sn[2 * k] = b2.x;
            sn[2 * k + 1] = b2.y;
        }
        wave_lds_fence();

        // ---- B: radial return
        double sig[D], nrm[D], dp, a, b;
        vm_return_map<D>(c, e, sn, p_l, sig, dp, nrm, a, b);

        // ---- C: as vm_tile
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            X2[lane * T::CH_VEC + k] = dxo_f64x2{sig[2 * k], sig[2 * k + 1]};
            if constexpr (MODE == 0) Y2[lane * (T::ST / 2) + k] = dxo_f64x2{nrm[2 * k], nrm[2 * k + 1]};     // the tangent walk's inputs: only where a tangent is written
        }
        if constexpr (MODE == 0) Y2[lane * (T::ST / 2) + T::CH_VEC] = dxo_f64x2{a, b};
        wave_lds_fence();
        if (lane < npts) store8<NT>(dp_out + p0 + lane, dp);
        dxo_f64x2* g_o = reinterpret_cast<dxo_f64x2*>(sigma + p0 * D);
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < nvec) store16<NT>(g_o + idx, X2[idx]);
        }
        if constexpr (RES) {
            // virtual work of the returned stress: T = w |J| sigma_hat J^-T per point, then f_a = sum_q T_q dphi_a(q) across the cell's lanes
            constexpr double r2 = 0.70710678118654752440;
            const double gh[3][3] = {{sig[0], r2 * sig[3], r2 * sig[4]}, {r2 * sig[3], sig[1], r2 * sig[5]}, {r2 * sig[4], r2 * sig[5], sig[2]}};
            const bool has_point = lane < npts;
            double sK[3][3], Tq[3][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) sK[k / 3][k % 3] = Y[k * DXO_WAVE + lane];
            sK[2][2] = sK22;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) tt += gh[i][j] * sK[k][j];
                    Tq[i][k] = has_point ? tt : 0.0;
                }
            if constexpr (DXO_VMF_RES_MFMA) {
                wave_lds_fence();           // the stress rows have left X, w|J|J^-1 has left Y: the whole slice stages T
                c8m_d4 acc[2][2];
                const double none[2][6] = {};
                c8m_contract<true>(W, lane, Tq, none, acc, tab8);
                c8m_store<ND_CT>(m, lane, acc, cell0 + c0, ncell, fe, nullptr);
                continue;
            }
            const C8Lane L8(tab8, lane);
            const int64_t cell = cell0 + c0 + (lane >> 3);
            c8_scatter<ND_CT>(L8, Tq, [&](int a, const double (&o)[3]) {
                if (!has_point) return;
#pragma unroll
                for (int i = 0; i < 3; ++i) fe[((int64_t)a * m.num_cells_fe + cell) * 3 + i] = o[i];
            });
            wave_lds_fence();
            continue;
        }
        if constexpr (MODE == 1) {      // (sigma, dp) only: a matrix-free solver rebuilds the tangent's action from them (dxo_tangent_apply_vm)
            wave_lds_fence();
            continue;
        }
#if DXO_VMF_ROWS
        wave_lds_fence();       // sigma has left X: the whole slice stages tangent rows now
        vm_store_tangent_rows<D, NT>(c, W, nrm, a, b, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
        continue;
#endif
#if DXO_VMF_FULL
        // a full group of 64 points (8 cells x 8 points on hexahedra): guard-free tangent stores, as in vm_tile
        if (npts == DXO_WAVE) vm_store_tangent<D, NT, true>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
        else
#endif
        vm_store_tangent<D, NT>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
        wave_lds_fence();
    }
}

struct FieldLaunch {
    VmConst c;
    const dxo_mesh* mesh;
    const double* d_u;
    int64_t next_cell;     // host pipeline: chunks arrive in order, each covers the next n_chunk cells
    // host half (option vm_host_tangent, see von_mises.hip): the caller's arrays, whole
    const double* h_sigma = nullptr;
    double* h_dp = nullptr;
    double* h_C_tang = nullptr;
};

int field_launch(dxo_ctx* ctx, const FieldLaunch& L, int64_t cell0, int64_t n_cells, const double* sigma_n,
                 const double* p, double* C_tang, double* sigma, double* dp, hipStream_t s, double* fe = nullptr) {
    if (n_cells == 0) return DXO_OK;
    const OperandDev& m = L.mesh->dev;
    const int D = L.mesh->gdim == 2 ? 4 : 6;
    int wd = m.cells_per_wave * (op_odd(m.ndofs * L.mesh->gdim) + op_odd(m.ngeom * L.mesh->gdim));
    int tile = DXO_WAVE * D + DXO_WAVE * (D + 2);
#if DXO_VMF_ROWS
    if (C_tang && tile < (D == 4 ? VM_ROWS_DOUBLES<4> : VM_ROWS_DOUBLES<6>)) tile = D == 4 ? VM_ROWS_DOUBLES<4> : VM_ROWS_DOUBLES<6>;
#endif
    if (wd < tile) wd = tile;
    wd = (wd + 1) & ~1;
    const int mode = fe ? 2 : (C_tang ? 0 : 1);
    const bool q2hex = DXO_OP_CT && L.mesh->gdim == 3 && m.ndofs == 27 && m.ngeom == 8;   // trip counts known at compile time
    // the landing slices of the global_load_lds prefetch (4 waves x 64 x (D + 1) doubles: 14 KB at d = 6) are requested only by the
    // instantiations that use them, and only when they fit beside the tables and the four gather regions; a larger element (Q2
    // hexahedra with the 27-point rule: 57 KB without them) takes the same kernel with the loads in registers
    const size_t base_doubles = (size_t)m.table_doubles + 4 * (size_t)wd + (fe ? C8_LDS : 0);
    const size_t slice_doubles = 4 * (size_t)DXO_WAVE * (D + 1);
    bool dma = vmf_dma_compiled(mode, (mode == 2 || q2hex) ? 27 : 0);
    if (dma && (base_doubles + slice_doubles) * sizeof(double) > 64 * 1024) {
        if (mode == 2) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_field: element too large for the LDS budget");   // cannot happen: the residual form is the 8-point rule's
        dma = false;
    }
    const size_t shm = (base_doubles + (dma ? slice_doubles : 0)) * sizeof(double);
    if (shm > 64 * 1024) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_field: element too large for the LDS budget");
    const int64_t n_groups = (n_cells + m.cells_per_wave - 1) / m.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * ((C_tang || fe) ? DXO_VMF_BLOCKS_PER_CU : DXO_VMF_STATE_BLOCKS_PER_CU);
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;      // whole rounds over the 8 XCDs (xcd_group_walk)
    const bool nt = ctx->nontemporal != 0;
#define DXO_VMF_LAUNCH(...)                                                                                                         \
    do {                                                                                                                            \
        hipLaunchKernelGGL((vm_field<__VA_ARGS__>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u,  \
                           sigma_n, p, C_tang, sigma, dp, L.mesh->d_wq, fe);                                                       \
    } while (0)
    if (mode == 2) {            // residual form (dxo_vmf_residual_eligible has been checked by the caller)
        if (nt) DXO_VMF_LAUNCH(3, true, 27, 8, 2); else DXO_VMF_LAUNCH(3, false, 27, 8, 2);
    } else if (q2hex) {
        if (mode == 0) { if (nt) DXO_VMF_LAUNCH(3, true, 27, 8, 0); else DXO_VMF_LAUNCH(3, false, 27, 8, 0); }
        else if (dma)  { if (nt) DXO_VMF_LAUNCH(3, true, 27, 8, 1); else DXO_VMF_LAUNCH(3, false, 27, 8, 1); }
        else           { if (nt) DXO_VMF_LAUNCH(3, true, 27, 8, 1, false); else DXO_VMF_LAUNCH(3, false, 27, 8, 1, false); }     // a larger rule than 2x2x2
    } else if (L.mesh->gdim == 2) {
        if (dma) {
            if (mode == 0) { if (nt) DXO_VMF_LAUNCH(2, true, 0, 0, 0); else DXO_VMF_LAUNCH(2, false, 0, 0, 0); }
            else           { if (nt) DXO_VMF_LAUNCH(2, true, 0, 0, 1); else DXO_VMF_LAUNCH(2, false, 0, 0, 1); }
        } else {
            if (mode == 0) { if (nt) DXO_VMF_LAUNCH(2, true, 0, 0, 0, false); else DXO_VMF_LAUNCH(2, false, 0, 0, 0, false); }
            else           { if (nt) DXO_VMF_LAUNCH(2, true, 0, 0, 1, false); else DXO_VMF_LAUNCH(2, false, 0, 0, 1, false); }
        }
    } else {
        if (dma) {
            if (mode == 0) { if (nt) DXO_VMF_LAUNCH(3, true, 0, 0, 0); else DXO_VMF_LAUNCH(3, false, 0, 0, 0); }
            else           { if (nt) DXO_VMF_LAUNCH(3, true, 0, 0, 1); else DXO_VMF_LAUNCH(3, false, 0, 0, 1); }
        } else {
            if (mode == 0) { if (nt) DXO_VMF_LAUNCH(3, true, 0, 0, 0, false); else DXO_VMF_LAUNCH(3, false, 0, 0, 0, false); }
            else           { if (nt) DXO_VMF_LAUNCH(3, true, 0, 0, 1, false); else DXO_VMF_LAUNCH(3, false, 0, 0, 1, false); }
        }
    }
#undef DXO_VMF_LAUNCH
    return DXO_OK;
}

int field_chunk(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const* d_in, void* const* d_out, hipStream_t s) {
    FieldLaunch& L = *static_cast<FieldLaunch*>(user);
    const int64_t cell0 = L.next_cell;
    L.next_cell += n_chunk;
    // host-rebuild mode: the tangent is rebuilt from (sigma, dp) on the host, the device writes none (the (sigma, dp)-only launch)
    return field_launch(ctx, L, cell0, n_chunk, (const double*)d_in[0], (const double*)d_in[1], L.h_C_tang ? nullptr : (double*)d_out[0],
                        (double*)d_out[1], (double*)d_out[2], s);
}

// post hook of the host pipeline: C_tang of the chunk's cells from the (sigma, dp) that have just landed
int field_host_rebuild(dxo_ctx* ctx, void* user, int64_t first_cell, int64_t n_cells) {
    const FieldLaunch& L = *static_cast<const FieldLaunch*>(user);
    const int nq = L.mesh->dev.nq, d = L.mesh->gdim == 2 ? 4 : 6;
    const int64_t first = first_cell * nq, m = n_cells * nq;
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

int field_upload_u(dxo_ctx* ctx, dxo_mesh* mesh, const double* u) {
    const size_t ub = (size_t)mesh->num_field_nodes * mesh->gdim * sizeof(double);
    if (mesh->u_cap < ub) {
        if (mesh->d_u) DXO_HIP(ctx, hipFree(mesh->d_u));
        mesh->d_u = nullptr;
        mesh->u_cap = 0;
        DXO_HIP(ctx, hipMalloc((void**)&mesh->d_u, ub));
        mesh->u_cap = ub;
    }
    DXO_HIP(ctx, hipMemcpy(mesh->d_u, u, ub, hipMemcpyHostToDevice));
    return DXO_OK;
}

}  // namespace

// Residual form for dxo_von_mises_residual (adjoint.hip owns the element-vector buffer and the node sums): Q2 hexahedra with the
// 2x2x2 rule whose gather fits the register pipeline.
bool dxo_vmf_residual_eligible(const dxo_mesh* mesh) {
    const OperandDev& m = mesh->dev;
    return DXO_OP_CT && mesh->gdim == 3 && m.ndofs == 27 && m.ngeom == 8 && m.nq == 8 && m.cells_per_wave == 8 &&
           m.cells_per_wave * m.ndofs <= OP_GI * DXO_WAVE && m.cells_per_wave * m.ngeom <= OP_XI * DXO_WAVE;
}

int dxo_vmf_residual_launch(dxo_ctx* ctx, const dxo_vm_params* prm, const dxo_mesh* mesh, const double* u, const double* sigma_n,
                            const double* p, double* sigma, double* dp, double* fe, hipStream_t s) {
    FieldLaunch L{make_const(*prm), mesh, u, 0};
    return field_launch(ctx, L, 0, mesh->num_cells, sigma_n, p, nullptr, sigma, dp, s, fe);
}

// Fused strain + return map with the history variables in a dxo_vm_state (von_mises.hip): of a host call only the dof
// vector goes up (one value per dof, not per quadrature point) and, with option vm_host_tangent, only (sigma, dp) come
// back — 56 instead of 448 B/point (d = 6) on the PCIe link.
extern "C" int dxo_von_mises_field_state(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, dxo_vm_state* st, int mem,
                                         const double* u, double* C_tang, double* sigma, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !mesh || !st) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field_state: NULL params, mesh or state");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_von_mises_field_state: bad mem");
    if (!st->uploaded) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_field_state: dxo_vm_state_upload has not been called");
    const int64_t nc = mesh->num_cells;
    const int G = mesh->gdim, D = G == 2 ? 4 : 6, nq = mesh->dev.nq;
    if (st->d != D || st->n != nc * nq)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_field_state: the state does not cover num_cells*nq points of this mesh's d");
    if (nc == 0) return DXO_OK;
    if (!u || (!C_tang && mem != DXO_MEM_DEVICE)) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field_state: NULL array (C_tang may be NULL on the device path only)");
    if (((uintptr_t)u | (uintptr_t)C_tang | (uintptr_t)sigma | (uintptr_t)dp) & 7u)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field_state: arrays must be 8-byte aligned");
    FieldLaunch L{make_const(*prm), mesh, u, 0};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    st->has_result = false;
    if (mem == DXO_MEM_DEVICE) {
        if ((uintptr_t)C_tang & 15u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field_state: device C_tang must be 16-byte aligned");
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = field_launch(ctx, L, 0, nc, st->sigma_n, st->p, C_tang, st->sigma, st->dp, s);
        if (rc != DXO_OK) return rc;
        if (sigma) DXO_HIP(ctx, hipMemcpyAsync(sigma, st->sigma, (size_t)st->n * D * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (dp) DXO_HIP(ctx, hipMemcpyAsync(dp, st->dp, (size_t)st->n * sizeof(double), hipMemcpyDeviceToDevice, s));
        rc = dxo_device_end(ctx, s);
        if (rc == DXO_OK) st->has_result = true;
        return rc;
    }
    if (!sigma || !dp) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field_state: host sigma / dp are required");
    int rc = field_upload_u(ctx, mesh, u);
    if (rc != DXO_OK) return rc;
    L.d_u = mesh->d_u;
    const size_t sd = sizeof(double) * (size_t)nq;
    std::vector<dxo_span> in = {{nullptr, nullptr, D * sd, st->sigma_n}, {nullptr, nullptr, sd, st->p}};
    if (ctx->vm_host_tangent && st->n >= ctx->vm_rebuild_min_points) {
        L.h_sigma = sigma;
        L.h_dp = dp;
        L.h_C_tang = C_tang;
        L.c.mark_indeterminate = 1;
        std::vector<dxo_span> out = {{nullptr, nullptr, 0}, {nullptr, sigma, D * sd, st->sigma}, {nullptr, dp, sd, st->dp}};
        const int64_t saved_chunk = ctx->host_chunk_points;
        if (ctx->host_chunk_points > ctx->vm_rebuild_chunk_points) ctx->host_chunk_points = ctx->vm_rebuild_chunk_points;
        rc = dxo_run_host_pipeline(ctx, nc, in, out, field_chunk, &L, nq, field_host_rebuild, true);
        ctx->host_chunk_points = saved_chunk;
    } else {
        std::vector<dxo_span> out = {{nullptr, C_tang, D * D * sd}, {nullptr, sigma, D * sd, st->sigma}, {nullptr, dp, sd, st->dp}};
        rc = dxo_run_host_pipeline(ctx, nc, in, out, field_chunk, &L, nq, nullptr, true);
    }
    if (rc == DXO_OK) st->has_result = true;
    return rc;
}

extern "C" int dxo_von_mises_field(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, int mem, const double* u,
                                   const double* sigma_n, const double* p, double* C_tang, double* sigma, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field: NULL params or mesh");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_von_mises_field: bad mem");
    const int64_t nc = mesh->num_cells;
    if (nc == 0) return DXO_OK;
    if (!u || !sigma_n || !p || (!C_tang && mem != DXO_MEM_DEVICE) || !sigma || !dp)
        return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field: NULL array (C_tang may be NULL on the device path only)");
    const uintptr_t all = (uintptr_t)u | (uintptr_t)sigma_n | (uintptr_t)p | (uintptr_t)C_tang | (uintptr_t)sigma | (uintptr_t)dp;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field: arrays must be 8-byte aligned");
    const int G = mesh->gdim, D = G == 2 ? 4 : 6, nq = mesh->dev.nq;
    FieldLaunch L{make_const(*prm), mesh, u, 0};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (mem == DXO_MEM_DEVICE) {
        if (((uintptr_t)sigma_n | (uintptr_t)C_tang | (uintptr_t)sigma) & 15u)
            return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field: device sigma_n, C_tang, sigma must be 16-byte aligned");
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = field_launch(ctx, L, 0, nc, sigma_n, p, C_tang, sigma, dp, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    // host arrays: the field vector goes up whole (it is small: one value per dof, not per quadrature point), the
    // state and the outputs stream through the chunked pipeline in units of CELLS
    int rc = field_upload_u(ctx, mesh, u);
    if (rc != DXO_OK) return rc;
    L.d_u = mesh->d_u;
    const size_t sd = sizeof(double) * (size_t)nq;
    std::vector<dxo_span> in = {{sigma_n, nullptr, D * sd}, {p, nullptr, sd}};
    if (ctx->vm_host_tangent && nc * nq >= ctx->vm_rebuild_min_points) {   // (sigma, dp) back, C_tang rebuilt on the host
        L.h_sigma = sigma;
        L.h_dp = dp;
        L.h_C_tang = C_tang;
        L.c.mark_indeterminate = 1;
        std::vector<dxo_span> out = {{nullptr, nullptr, 0}, {nullptr, sigma, D * sd}, {nullptr, dp, sd}};
        const int64_t saved_chunk = ctx->host_chunk_points;
        if (ctx->host_chunk_points > ctx->vm_rebuild_chunk_points) ctx->host_chunk_points = ctx->vm_rebuild_chunk_points;
        rc = dxo_run_host_pipeline(ctx, nc, in, out, field_chunk, &L, nq, field_host_rebuild, true);
        ctx->host_chunk_points = saved_chunk;
        return rc;
    }
    std::vector<dxo_span> out = {{nullptr, C_tang, D * D * sd}, {nullptr, sigma, D * sd}, {nullptr, dp, sd}};
    return dxo_run_host_pipeline(ctx, nc, in, out, field_chunk, &L, nq, nullptr, true);
}
