// Experiment (not shipped; built with -DDXO_EXPERIMENTS -DDXO_C8M_FORWARD=1): the strain contraction of tangent_apply<3, 27, 8, ..., MF> on the
// f64 matrix pipe as well. Correct (4e-16 from the DPP / LDS form, bit-reproducible) and no faster: with the scatter's 24 MFMAs the forward's
// 28 make the matrix pipe the longest of the wave's queues (52 x 64 cycles per group of 64 points; v_mfma_f64_16x16x4_f64 has the vector
// pipe's fp64 rate on gfx950 and a quarter of the tiles are padding). Q2 hexahedra 108^3, ms per call, same lease:
//   state-based action: DPP 0.874-0.890, MFMA scatter 0.813, MFMA scatter + forward 0.822;  C_tang rows: 1.13 / 1.10 / 1.11
// Included by csrc/adjoint.hip ahead of tangent_apply.
#pragma once

// ---- the FORWARD contraction the same way: gref_(c,q)[i][k] = sum_a u_(c,a)[i] dphi_a,k(xi_q) = D[(q, k)][(c, i)] = F[(q, k)][a] U[a][(c, i)],
// 28 MFMAs (rows 24 of 32, columns 24 of 32, 27 of 28 nodes) instead of 162 LDS reads + 243 FMAs per lane.
constexpr int C8M_FTAB = 14 * DXO_WAVE;    // doubles: F fragments [mt * 7 + s][lane], lane l: (q, k) = mt * 16 + l % 16, node 4 s + l / 16
constexpr int C8M_ZS = 25;                 // column stride of the staged result (odd)

template <int ND>
__device__ __forceinline__ void c8m_fill_F(const OperandDev& m, double* Ftab) {
    for (int e = threadIdx.x; e < C8M_FTAB; e += blockDim.x) {
        const int f = e / DXO_WAVE, lane = e - f * DXO_WAVE, mt = f / 7, s = f - mt * 7;
        const int r = mt * 16 + (lane & 15), a = 4 * s + (lane >> 4);
        Ftab[e] = (r < 24 && a < ND) ? m.dphi[((r / 3) * m.ndofs + a) * 3 + r % 3] : 0.0;
    }
}

// operand_compute_geo<3, 3, DXO_OPERAND_EPS_MANDEL, ND, 8> with the contraction on the matrix pipe. W: the gather buffer (dofs of the
// wave's 8 cells, then their vertices); the result passes through the dof part of W on its way from the MFMA layout to lane = point.
template <int ND>
__device__ __forceinline__ bool c8m_forward_eps(const OperandDev& m, const double* tab, const double* Ftab, double* W, int ncell, int lane,
                                                double (&e)[6], double (&K)[3][3], double& detJ) {
    constexpr int su = (ND * 3) | 1;                  // op_odd
    const OperandLayout<3> L(m);
    double* U = W;
    const double* X = W + 8 * su;
    op_fence();
    const int c = lane >> 3, q = lane & 7;
    const bool active = c < ncell;
    c8m_d4 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = c8m_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 7; ++st) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 16 + (lane & 15), a = 4 * st + (lane >> 4), cc = n / 3;
            const double b = (n < 24 && a < ND) ? U[cc * su + a * 3 + (n - 3 * cc)] : 0.0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ftab[(mt * 7 + st) * DXO_WAVE + lane], b, acc[mt][nt], 0, 0, 0);
        }
    }
    // geometry of this lane's point while the matrix pipe works
    if (active) {
        const double* dpsi = tab + L.o_dpsi + q * L.sdpsi;
        const double* Xc = X + c * L.sx;
        double J[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) J[j][k] = 0.0;
#pragma unroll
        for (int v = 0; v < 8; ++v)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) J[j][k] += Xc[v * 3 + j] * dpsi[v * 3 + k];
        detJ = invert<3>(J, K);
    }
    op_fence();                                   // every lane has read its dofs: the dof part of W becomes the staging area
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 16 + (lane & 15);
        if (n < 24) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + 4 * r + (lane >> 4);
                    if (mt == 0 || r < 2) U[n * C8M_ZS + row] = acc[mt][nt][r];       // rows 24..31 are padding
                }
        }
    }
    op_fence();
    if (active) {
        double g[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double gr[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) gr[k] = U[(3 * c + i) * C8M_ZS + 3 * q + k];
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i][j] = gr[0] * K[0][j] + gr[1] * K[1][j] + gr[2] * K[2][j];
        }
        const double val[3] = {0.0, 0.0, 0.0};
        shape_operand<3, 3, DXO_OPERAND_EPS_MANDEL>(val, g, e);
    }
    op_fence();   // W may be reused by the caller
    return active;
}
