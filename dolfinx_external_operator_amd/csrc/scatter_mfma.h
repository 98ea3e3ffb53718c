// scatter_mfma.h — cell8_mfma.h's scatter for the other standard elements: the element vectors of a wave's cells as one small product on
// the fp64 matrix pipe (option adjoint_mfma),
//   f[a][(c, i)] = sum over (q, k) of dphi_a,k(xi_q) * T_(c,q)[i][k]      a: ND nodes, (c, i): CPW cells x G, (q, k): NQ points x G,
// MT x NT tiles of v_mfma_f64_16x16x4_f64 over KS steps of four rows: P2 tetrahedra (10 nodes, 4 points, 16 cells per wave) 1 x 3 x 3 = 9
// instructions per group, P2 triangles (6, 3, 21) 1 x 3 x 2 = 6 — instead of adjoint_scatter's lane = (cell, node) loop over the parked
// tensors (96 LDS reads per (cell, node) pair). The reference adds element vectors cell by cell (DOLFINx assembly of the forms of
// external_operator.py:463-486). Same sums in another fixed order: equal to adjoint_scatter to rounding, bit-reproducible.
#pragma once
#include "cell8_mfma.h"

namespace {

// RPQ: rows per quadrature point — G (the directions k of the scatter) or 3 (a pass of three (k, kk) pairs of the diagonal)
template <int G, int ND, int NQ, int RPQ = G>
struct GmShape {
    static constexpr int CPW = DXO_WAVE / NQ;                 // cells per wave (OperandDev::cells_per_wave of such a mesh)
    static constexpr int ROWS = NQ * RPQ, COLS = CPW * G;
    static constexpr int MT = (ND + 15) / 16, KS = (ROWS + 3) / 4, NT = (COLS + 15) / 16;
    // column stride of the staged T: even with an odd half, so the 16 columns x 2 rows a 32-lane half reads fall on 32 different 8-byte banks
    static constexpr int CS = ROWS <= 18 ? 18 : 34;
    static constexpr int ATAB = MT * KS * DXO_WAVE;           // doubles: A fragments [mt * KS + s][lane]
    static constexpr int STAGE = COLS * CS;                   // doubles of the wave's staging slice
    static_assert(ROWS <= CS && MT <= 2, "shape outside the staged layout");
};

// A fragments: lane l of fragment (mt, s) holds dphi of node mt * 16 + l % 16 at row r = 4 s + l / 16 = q * G + k (0 beyond the element)
template <int G, int ND, int NQ>
__device__ __forceinline__ void gm_fill_A(const OperandDev& m, double* Atab) {
    using S = GmShape<G, ND, NQ>;
    for (int e = threadIdx.x; e < S::ATAB; e += blockDim.x) {
        const int f = e / DXO_WAVE, lane = e - f * DXO_WAVE, mt = f / S::KS, s = f - mt * S::KS;
        const int a = mt * 16 + (lane & 15), r = 4 * s + (lane >> 4);
        Atab[e] = (a < ND && r < S::ROWS) ? m.dphi[((r / G) * m.ndofs + a) * G + r % G] : 0.0;
    }
}

// tangent_diag: rows (q, pair) with the products dphi_a,k dphi_a,kk of the pairs (k <= kk) 3 p .. 3 p + 2 of (00, 01, 02, 11, 12, 22) resp.
// (00, 01, 11); PASSES tables of GmShape<G, ND, NQ, 3>::ATAB doubles
template <int G, int ND, int NQ>
__device__ __forceinline__ void gm_fill_A2(const OperandDev& m, double* Atab) {
    using S = GmShape<G, ND, NQ, 3>;
    constexpr int PASSES = G == 3 ? 2 : 1;
    for (int e = threadIdx.x; e < PASSES * S::ATAB; e += blockDim.x) {
        const int p = e / S::ATAB, e1 = e - p * S::ATAB;
        const int f = e1 / DXO_WAVE, lane = e1 - f * DXO_WAVE, mt = f / S::KS, s = f - mt * S::KS;
        const int a = mt * 16 + (lane & 15), r = 4 * s + (lane >> 4), q = r / 3, slot = 3 * p + r % 3;
        int k, kk;
        if (G == 3) { k = slot < 3 ? 0 : (slot < 5 ? 1 : 2); kk = slot < 3 ? slot : (slot < 5 ? slot - 2 : 2); }
        else { k = slot < 2 ? 0 : 1; kk = slot < 1 ? 0 : 1; }
        double v = 0.0;
        if (a < ND && r < S::ROWS) {
            const double* d = m.dphi + (q * m.ndofs + a) * G;
            v = d[k] * d[kk];
        }
        Atab[e] = v;
    }
}

// acc (+)= A x T for the wave's cells. Tl: the wave's staging slice (S::STAGE doubles); T[i][r]: this lane's point, component i, row r of the
// point's RPQ (zero for lanes without a point).
template <int G, int ND, int NQ, int RPQ, bool ACCUM>
__device__ __forceinline__ void gm_contract(const double* Atab, double* Tl, int lane, const double (&T)[G][RPQ],
                                            c8m_d4 (&acc)[GmShape<G, ND, NQ, RPQ>::MT][GmShape<G, ND, NQ, RPQ>::NT]) {
    using S = GmShape<G, ND, NQ, RPQ>;
    const int c_l = lane / NQ, q_l = lane - c_l * NQ;
    if (c_l < S::CPW) {
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < RPQ; ++k) Tl[(c_l * G + i) * S::CS + q_l * RPQ + k] = T[i][k];
    }
    op_fence();
    if constexpr (!ACCUM) {
#pragma unroll
        for (int mt = 0; mt < S::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < S::NT; ++nt) acc[mt][nt] = c8m_d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int st = 0; st < S::KS; ++st) {
        const int r = 4 * st + (lane >> 4);
#pragma unroll
        for (int nt = 0; nt < S::NT; ++nt) {
            const int n = nt * 16 + (lane & 15);
            const double b = (n < S::COLS && r < S::ROWS) ? Tl[n * S::CS + r] : 0.0;
#pragma unroll
            for (int mt = 0; mt < S::MT; ++mt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(Atab[(mt * S::KS + st) * DXO_WAVE + lane], b, acc[mt][nt], 0, 0, 0);
        }
    }
    op_fence();                                   // the slice is free again
}

// acc[mt][nt][r] of lane l = f[mt * 16 + 4 r + l / 16][nt * 16 + l % 16] (D layout of the instruction, scripts/exp/mfma64_probe.hip)
template <int G, int ND, int NQ>
__device__ __forceinline__ void gm_store(const OperandDev& m, int lane, const c8m_d4 (&acc)[GmShape<G, ND, NQ>::MT][GmShape<G, ND, NQ>::NT], int64_t c0,
                                         int ncell, double* __restrict__ fe, double* __restrict__ out) {
    using S = GmShape<G, ND, NQ>;
#pragma unroll
    for (int nt = 0; nt < S::NT; ++nt) {
        const int n = nt * 16 + (lane & 15), c = n / G, i = n - G * c;
        if (n < S::COLS && c < ncell) {
            const int64_t cell = c0 + c;
#pragma unroll
            for (int mt = 0; mt < S::MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = mt * 16 + 4 * r + (lane >> 4);
                    if (a < ND) {
                        if (fe) fe[((int64_t)a * m.num_cells_fe + cell) * G + i] = acc[mt][nt][r];
                        else unsafeAtomicAdd(out + (int64_t)m.dofmap[cell * ND + a] * G + i, acc[mt][nt][r]);
                    }
                }
        }
    }
}

// Tl: the wave's staging slice (S::STAGE doubles); T = w |det J| G_hat K^T of this lane's point (zero for lanes without one).
template <int G, int ND, int NQ>
__device__ __forceinline__ void gm_scatter(const OperandDev& m, const double* Atab, double* Tl, int lane, const double (&T)[G][G], int64_t c0, int ncell,
                                           double* __restrict__ fe, double* __restrict__ out) {
    using S = GmShape<G, ND, NQ>;
    c8m_d4 acc[S::MT][S::NT];
    gm_contract<G, ND, NQ, G, false>(Atab, Tl, lane, T, acc);
    gm_store<G, ND, NQ>(m, lane, acc, c0, ncell, fe, out);
}

}  // namespace
