// cell8_mfma.h — the consumer-side scatter of hexahedra with the 2x2x2 rule on the fp64 MATRIX pipe (option adjoint_mfma).
// Used by adjoint.hip (operand_adjoint_c8_mfma, tangent_apply<3, 27, 8, .., MF>, tangent_diag<3, 27, VM, MF>) and vm_field.hip (residual form).
// The reference adds element vectors cell by cell (DOLFINx assembly of the forms of external_operator.py:463-486); the product form of
// that sum for a wave's 8 cells is below.
#pragma once
#include "operand_core.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// The scatter
//   f[a][(c, i)] = sum over (q, k) of dphi_a,k(xi_q) * T_(c,q)[i][k]          a: 27 nodes, (c, i): 8 cells x 3, (q, k): 8 points x 3
// is a 27 x 24 x 24 product per wave group: 24 v_mfma_f64_16x16x4_f64 (two 16-row tiles of nodes, two 16-column tiles of (cell,
// component), six K-steps) instead of 288 FMAs and 250 DPP moves / adds per lane. The table fragments A[a][(q, k)] are constants
// of the kernel (12 doubles per lane in registers, or a lane-linear LDS table); a lane's T goes through the wave's LDS slice once (9 writes,
// 12 fragment reads) to reach the B layout (lane l: row 4 s + l / 16, column l % 16). The matrix pipe gives no more flops than the
// vector pipe on gfx950 (a 16x16x4 f64 MFMA takes ~64 cycles for 1 024 FMAs) — what it gives is ISSUE SLOTS: the kernel was bound by
// the issue of ~900 vector instructions per group, a third of its lanes' work being data movement.
// Results: the same sums in another (fixed) order — equal to the DPP form to rounding, bit-reproducible run to run.
constexpr int C8M_CS = 34;                 // column stride of the staged T (doubles): column-major, 34 = 2 mod 32, so the 16 columns x 2 rows a 32-lane
                                           // half reads land on 32 different 8-byte banks
constexpr int C8M_WAVE = 24 * C8M_CS;      // doubles per wave (816: fits the gather buffer of tangent_apply<3, 27, 8>, 848)
typedef double c8m_d4 __attribute__((ext_vector_type(4)));

// A fragments of the scatter: lane l holds dphi of node mt * 16 + l % 16 at (point, direction) index r = 4 s + l / 16 (r = 3 q + k)
template <int ND>
__device__ __forceinline__ void c8m_load_A(const OperandDev& m, int lane, double (&Afr)[2][6]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const int a = mt * 16 + (lane & 15), r = 4 * s + (lane >> 4);
            Afr[mt][s] = a < ND ? m.dphi[((r / 3) * m.ndofs + a) * 3 + r % 3] : 0.0;
        }
}

// the same fragments as a lane-linear LDS table [mt * 6 + s][lane] (12 x 64 doubles, filled by the whole workgroup): 24 registers less
template <int ND>
__device__ __forceinline__ void c8m_fill_A(const OperandDev& m, double* Atab) {
    for (int e = threadIdx.x; e < 12 * DXO_WAVE; e += blockDim.x) {
        const int f = e / DXO_WAVE, lane = e - f * DXO_WAVE, mt = f / 6, s = f - mt * 6;
        const int a = mt * 16 + (lane & 15), r = 4 * s + (lane >> 4);
        Atab[e] = a < ND ? m.dphi[((r / 3) * m.ndofs + a) * 3 + r % 3] : 0.0;
    }
}

// tangent_diag: K_(a,i),(a,i) = sum_q sum_(k <= kk) dphi_a,k dphi_a,kk N_(c,q)[i][(k, kk)] — the same product with 48 rows (q, (k, kk)),
// as two passes of 24: pass p covers the pairs 3 p .. 3 p + 2 of (00, 01, 02, 11, 12, 22). Two tables [p][mt * 6 + s][lane].
template <int ND>
__device__ __forceinline__ void c8m_fill_A2(const OperandDev& m, double* Atab) {
    for (int e = threadIdx.x; e < 2 * 12 * DXO_WAVE; e += blockDim.x) {
        const int p = e / (12 * DXO_WAVE), e1 = e - p * 12 * DXO_WAVE;
        const int f = e1 / DXO_WAVE, lane = e1 - f * DXO_WAVE, mt = f / 6, s = f - mt * 6;
        const int a = mt * 16 + (lane & 15), r = 4 * s + (lane >> 4), q = r / 3, slot = 3 * p + r % 3;
        const int k = slot < 3 ? 0 : (slot < 5 ? 1 : 2), kk = slot < 3 ? slot : (slot < 5 ? slot - 2 : 2);
        const double* d = m.dphi + (q * m.ndofs + a) * 3;
        Atab[e] = a < ND ? d[k] * d[kk] : 0.0;
    }
}

// f[a][(c, i)] = sum over (q, k) of dphi_a,k(xi_q) T_(c,q)[i][k] for the wave's 8 cells as 24 f64 MFMAs. `Tl`: the wave's staging slice
// (C8M_WAVE doubles: column (c, i) at n * C8M_CS, row (q, k) inside it). Every lane ends up with 16 entries: acc[mt][nt][r] belongs to node
// mt * 16 + 4 r + l / 16 and column n = nt * 16 + l % 16 = 3 c + i (D layout of v_mfma_f64_16x16x4_f64: scripts/exp/mfma64_probe.hip).
// ALDS: the A fragments come from the table of c8m_fill_A (`Atab`) instead of from `Afr`. ACCUM: `acc` is added to, not cleared
// (tangent_diag: 48 rows as two passes of 24).
template <bool ALDS = false, bool ACCUM = false>
__device__ __forceinline__ void c8m_contract(double* Tl, int lane, const double (&T)[3][3], const double (&Afr)[2][6], c8m_d4 (&acc)[2][2],
                                             const double* Atab = nullptr) {
    const int c_l = lane >> 3, q_l = lane & 7;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) Tl[(c_l * 3 + i) * C8M_CS + q_l * 3 + k] = T[i][k];
    op_fence();
    if constexpr (!ACCUM) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = c8m_d4{0.0, 0.0, 0.0, 0.0};
    }
#pragma unroll
    for (int st = 0; st < 6; ++st) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = nt * 16 + (lane & 15);
            const double b = n < 24 ? Tl[n * C8M_CS + 4 * st + (lane >> 4)] : 0.0;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                double a;
                if constexpr (ALDS) a = Atab[(mt * 6 + st) * DXO_WAVE + lane];
                else a = Afr[mt][st];
                acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[mt][nt], 0, 0, 0);
            }
        }
    }
    op_fence();                                   // the slice is free again
}

// the lane's 16 entries to the element-vector array fe[node][cell][component] (or, fe == nullptr, added to `out` with atomics)
template <int ND>
__device__ __forceinline__ void c8m_store(const OperandDev& m, int lane, const c8m_d4 (&acc)[2][2], int64_t c0, int ncell, double* __restrict__ fe,
                                          double* __restrict__ out) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 16 + (lane & 15), c = n / 3, i = n - 3 * c;
        if (n < 24 && c < ncell) {
            const int64_t cell = c0 + c;
            if (fe) {
                // node a = mt * 16 + 4 r + l / 16: a running pointer instead of sixteen 64-bit index products
                const int64_t step = m.num_cells_fe * 3;
                double* p = fe + ((int64_t)(lane >> 4) * m.num_cells_fe + cell) * 3 + i;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = mt * 16 + 4 * r + (lane >> 4);
                        if (a < ND) p[(int64_t)(mt * 16 + 4 * r) * step] = acc[mt][nt][r];
                    }
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a = mt * 16 + 4 * r + (lane >> 4);
                        if (a < ND) unsafeAtomicAdd(out + (int64_t)m.dofmap[cell * ND + a] * 3 + i, acc[mt][nt][r]);
                    }
            }
        }
    }
}

}  // namespace
