// cell8_brick.h — the consumer-side scatter of hexahedra with the 2x2x2 rule, BRICK form (round 6; option adjoint_brick).
//
// The reference adds element vectors into the dof vector cell by cell (DOLFINx assembly of the forms of
// src/dolfinx_external_operator/external_operator.py:463-486, driven by petsc/petsc.py:62-68). The two-pass form of adjoint.hip writes
// every (cell, local node) entry to `fe` (27 x 24 bytes per Q2 cell) and node_sum reads them back: x2.05-2.29 of the calls'
// algorithmic bytes (round 5). Here the library keeps its OWN cell order for these kernels — cells sorted along a Morton curve of
// their centroids, eight consecutive ones per wave group: on a structured mesh a 2 x 2 x 2 brick — and a group's eight element
// vectors meet inside the wave: the MFMA result goes through the wave's LDS slice once, entries of the same node are added there in
// a fixed order (ascending cell of the group), and the group writes ONE partial per node it touches — 125 instead of 216 on Q2
// bricks, 27 instead of 64 on Q1 — as one contiguous run. node_sum (unchanged kernel, other index arrays) adds at most 8 group
// partials per node instead of up to 27 entries. Any grouping is CORRECT (a group of unrelated cells just shares nothing); the Morton
// order only decides how much is shared. Per-point arrays stay in the caller's cell order: the kernels reach them through `orig`.
// Sums in another (fixed) order than the two-pass form: equal to rounding, bit-reproducible run to run.
#pragma once
#include "cell8_brick_host.h"
#include "cell8_mfma.h"
#include "operand_core.h"

namespace {

constexpr int BRICK_TAB_OFF = 27 * BRICK_DS + 1;      // doubles: where the group's reduction table sits behind D (676: 8-byte aligned bytes)
static_assert(BRICK_TAB_OFF * 8 + BRICK_TAB_BYTES <= C8M_WAVE * 8, "D + table must fit the wave's staging slice");
static_assert(BRICK_TAB_BYTES == BRICK_TAB_BYTES_H && BRICK_TAB_START == BRICK_TAB_START_H && BRICK_TAB_SRC == BRICK_TAB_SRC_H, "table layout: host and device");

// ---------------------------------------------------------------------------------------------------------------- device side
// The wave's 27 x 24 MFMA result (c8m_contract: acc[mt][nt][r] = D[node mt * 16 + 4 r + l / 16][column nt * 16 + l % 16], column = 3 cell + i)
// -> LDS -> one partial per (slot, component) -> part[(slot0 + u) * 3 + i], contiguous over the wave. `Tl`: the wave's staging slice
// (free again after c8m_contract's last fence); `tabw`: this lane's three dwords of the group's table, requested by the caller at the
// top of the iteration (brick_request_table).
struct BrickTabReg { uint32_t w[3]; };

__device__ __forceinline__ BrickTabReg brick_request_table(const BrickDev& br, int64_t grp, int lane) {
    BrickTabReg t;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(br.tabs + (int64_t)br.tab_id[grp] * BRICK_TAB_BYTES);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int w = k * DXO_WAVE + lane;
        t.w[k] = w < BRICK_TAB_BYTES / 4 ? src[w] : 0u;
    }
    return t;
}

template <int ND>
__device__ __forceinline__ void c8m_store_brick(double* Tl, int lane, const c8m_d4 (&acc)[2][2], const BrickTabReg& tabw, int64_t slot0,
                                                double* __restrict__ part) {
    // D and the table into the slice
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 16 + (lane & 15);
        if (n < 24) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a = mt * 16 + 4 * r + (lane >> 4);
                    if (a < ND) Tl[a * BRICK_DS + n] = acc[mt][nt][r];
                }
        }
    }
    uint32_t* tw = reinterpret_cast<uint32_t*>(Tl + BRICK_TAB_OFF);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int w = k * DXO_WAVE + lane;
        if (w < BRICK_TAB_BYTES / 4) tw[w] = tabw.w[k];
    }
    op_fence();
    const uint8_t* tb = reinterpret_cast<const uint8_t*>(tw);
    const int U3 = 3 * (int)tb[0];
    const uint16_t* src = reinterpret_cast<const uint16_t*>(tb + BRICK_TAB_SRC);
    double* dst = part + slot0 * 3;
    for (int o = lane; o < U3; o += DXO_WAVE) {
        const int u = o / 3, i = o - 3 * u;
        const int s0 = tb[BRICK_TAB_START + u], s1 = tb[BRICK_TAB_START + u + 1];
        double sum = 0.0;
        for (int s = s0; s < s1; ++s) sum += Tl[src[s] + i];
        dst[o] = sum;
    }
    op_fence();        // the slice is free again
}

}  // namespace
