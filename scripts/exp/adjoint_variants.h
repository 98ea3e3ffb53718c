// adjoint_variants.h — consumer-side kernel variants that were MEASURED AND NOT SHIPPED (profiles/r04_adjoint_experiments.txt).
// Included by csrc/adjoint.hip only when it is compiled with -DDXO_EXPERIMENTS (scripts/exp/build_variant.py <name> adjoint.hip
// -DDXO_EXPERIMENTS -DDXO_TA_C8_FORWARD=1); the product library does not contain them. The lane = cell tangent action
// (tangent_cell, -DDXO_TANGENT_CELL=1) sits in csrc/adjoint_cell.h under the same switch.
#pragma once

// EXPERIMENT, not launched by default (-DDXO_TA_C8_FORWARD=1): the same operator with the strain contraction ALSO across the lanes
// — NO gather buffer in LDS, dof values and vertex coordinates go from global memory into the registers of the lane that owns
// them and are all-gathered over the cell's 8 lanes (cell8_dpp.h). Correct (tests/test_adjoint_gpu.py passes on it) but slower
// than the shipped hybrid (LDS gather + contraction, DPP scatter): 1.65 against 1.31 ms per 10^7 points — the all-gather is 210
// dependent DPP moves per lane and group that the two resident waves per SIMD do not cover, while LDS reads are asynchronous;
// it also needs the tangent rows requested late (they do not fit beside the contraction's registers: 23 spilled otherwise).
template <int ND>
__global__ __launch_bounds__(DXO_BLOCK, 2) void tangent_apply_c8(OperandDev m, const double* __restrict__ wq,
                                                                 const double* __restrict__ C_tang, const double* __restrict__ v,
                                                                 int64_t n_cells, double* __restrict__ out, double* __restrict__ fe) {
    constexpr int D = 6;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    c8_fill_tables(m, lds);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + C8_LDS + wave * TangentRows<D>::LDS_DOUBLES;
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
    C8Pipe pf;
    int64_t grp = walk.first;
    c8_load_indices<ND>(m, pf, L, grp * cpw, cells_in(grp), lane);
    c8_load_values(m, pf, v);
    c8_load_indices<ND>(m, pf, L, (grp + stride) * cpw, cells_in(grp + stride), lane);
    const double w_l = wq[lane & 7];
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = cells_in(grp);
        const bool has_point = (lane >> 3) < ncell;
        TangentRows<D> rows;
#if DXO_C8_EARLY_C
        rows.request(C_tang, c0 * 8, ncell * 8, lane);
#endif
        // pf.ud / pf.xd hold THIS group's values (requested during the last iteration), pf.un / pf.xn the next group's indices
        double K[3][3], gref[3][3];
        const double det = c8_geometry(L, pf.xd, K);
        c8_forward(L, pf.ud, gref);
        // consumed in place: now the next group's values (indices have been here for an iteration) and the indices of the one after;
        // their latency runs under the tangent product and the scatter phase of this group and the geometry of the next
        c8_load_values(m, pf, v);
        c8_load_indices<ND>(m, pf, L, (grp + 2 * stride) * cpw, cells_in(grp + 2 * stride), lane);
        double e[D];
        {
            double val[3] = {0.0, 0.0, 0.0}, g[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double sacc = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) sacc += gref[i][k] * K[k][j];
                    g[i][j] = sacc;
                }
            shape_operand<3, 3, DXO_OPERAND_EPS_MANDEL>(val, g, e);
        }
        if (!has_point) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[k] = 0.0;
        }
#if !DXO_C8_EARLY_C
        __builtin_amdgcn_sched_barrier(0);     // keep the 18 row loads (72 registers) below the contraction
        rows.request(C_tang, c0 * 8, ncell * 8, lane);
#endif
        double t6[D];
        rows.times(W, lane, e, t6);
        double T[3][3];
        {
            double vh[3], gh[3][3];
            dual_tensor<3, 3, DXO_OPERAND_EPS_MANDEL>(t6, vh, gh);
            const double scale = w_l * fabs(det);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) tt += gh[i][j] * K[k][j];
                    T[i][k] = has_point ? scale * tt : 0.0;      // lanes without a point: zero vertices, singular J
                }
        }
        const int64_t cell = c0 + (lane >> 3);
        c8_scatter<ND>(L, T, [&](int a, const double (&o)[3]) {
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

