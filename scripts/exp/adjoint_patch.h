// adjoint_patch.h — the consumer-side scatter without the element-vector round trip through HBM.
//
// The two-pass form (adjoint.hip) writes every (cell, local node) entry of the element vectors to `fe` and a second kernel sums
// each node's entries in a fixed order: bit-reproducible, but on Q2 hexahedra 27 x 24 bytes per cell go out and come back —
// more than the call's algorithmic traffic (round 4: x2.05 of it by the counters). The reference's counterpart is DOLFINx
// assembly of inner(sigma, eps(v)) dx / the action of the Jacobian form (src/dolfinx_external_operator/external_operator.py:463-486),
// which adds element vectors into the dof vector cell by cell.
//
// Patch form: the wave groups (cells_per_wave consecutive cells, the unit the kernels work on) are cut into PATCHES of
// PATCH_WAVES x R groups by recursive coordinate bisection of their centroids (compact boxes of an exact group count). One
// workgroup of PATCH_WAVES waves owns a patch:
//   * wave w takes R groups of the patch, one after the other, and adds their element-vector entries into ITS OWN accumulator in
//     LDS (the nodes of its groups only) with plain read-add-write passes — within a group the cells are coloured at set-up so
//     that the lanes of one pass touch different nodes; no atomics, no barrier, nobody else writes there;
//   * after one barrier all threads MERGE: a patch node's value is the sum of its (at most PATCH_WAVES) per-wave partials in wave
//     order. A node all of whose cells lie in the patch goes straight to the dof vector, a node shared with another patch leaves
//     ONE partial per patch in `bpart`, and a second kernel adds the 2-8 partials of every such node in patch order
//     (node_sum_patch).
// Every addition happens in an order fixed at set-up: bit-reproducible run to run. Per cell of a 8x3x4-cell box of Q2 hexahedra
// ~6 partials instead of 27 element-vector entries travel through HBM.
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "dxo_common.h"
#include "operand_core.h"

#ifndef DXO_PATCH_R
#define DXO_PATCH_R 3                // wave groups per wave and patch (8 x 3 x 4 cells per patch on hexahedra, cells_per_wave = 8)
#endif
#ifndef DXO_PATCH_ACC_BYTES
#define DXO_PATCH_ACC_BYTES (40 * 1024)   // LDS accumulators of one patch (all waves): three workgroups per CU beside their tables
#endif

namespace {

// Host side, once per mesh: patches, per-wave numbering, merge maps, second-pass lists. `cell_xyz`: a representative point per
// cell (3 floats), kept by dxo_mesh_create. Returns false when the mesh cannot use the patch form (a single group's nodes do not
// fit a wave's accumulator, index overflow): the caller keeps the two-pass form.
struct PatchHost {
    std::vector<int32_t> groups;      // [n_patches][PATCH_WAVES][R]
    std::vector<int32_t> node_off;    // [n_patches + 1]
    std::vector<uint32_t> gnode;      // [n_slots]
    std::vector<uint16_t> mmap;       // [n_slots][PATCH_WAVES] index in wave w's accumulator, 0xffff: none
    std::vector<uint16_t> lnode;      // [num_cells][ndofs] index in the accumulator of the wave that owns the cell's group
    std::vector<uint8_t> cellcol, grp_ncol;
    std::vector<int32_t> bnode;
    std::vector<int64_t> bptr;
    std::vector<uint32_t> bent;
    int R = 0, max_priv = 0, max_local = 0;
    double shared_fraction = 0.0;
};

inline bool patch_build_host(const std::vector<int32_t>& dofmap, const std::vector<float>& cell_xyz, int64_t nc, int nd, int64_t nn, int cpw,
                             int bs_cap, int R, PatchHost& H) {
    const int64_t ng = (nc + cpw - 1) / cpw;
    const int64_t priv_cap = DXO_PATCH_ACC_BYTES / PATCH_WAVES / (int64_t)(bs_cap * sizeof(double));
    if (nc == 0 || R < 1 || (int64_t)cpw * nd > priv_cap || priv_cap >= 65535 || nn >= ((int64_t)1 << 31)) return false;
    const int T = PATCH_WAVES * R;
    // ---- group centroids, recursive bisection into boxes of exactly T groups (the last one may be short)
    std::vector<float> gx((size_t)ng * 3, 0.f);
    for (int64_t g = 0; g < ng; ++g) {
        const int64_t c1 = std::min(nc, (g + 1) * cpw);
        for (int64_t c = g * cpw; c < c1; ++c)
            for (int j = 0; j < 3; ++j) gx[(size_t)g * 3 + j] += cell_xyz[(size_t)c * 3 + j] / (float)(c1 - g * cpw);
    }
    std::vector<int32_t> order((size_t)ng);
    std::iota(order.begin(), order.end(), 0);
    auto sort_range = [&](int64_t lo, int64_t hi, int axis) {
        const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
        std::sort(order.begin() + lo, order.begin() + hi, [&](int32_t p, int32_t q) {
            const float* P = &gx[(size_t)p * 3];
            const float* Q = &gx[(size_t)q * 3];
            if (P[axis] != Q[axis]) return P[axis] < Q[axis];
            if (P[a2] != Q[a2]) return P[a2] < Q[a2];
            if (P[a1] != Q[a1]) return P[a1] < Q[a1];
            return p < q;
        });
    };
    std::vector<std::pair<int64_t, int64_t>> stack{{0, ng}};
    while (!stack.empty()) {
        const auto [lo, hi] = stack.back();
        stack.pop_back();
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int64_t k = lo; k < hi; ++k)
            for (int j = 0; j < 3; ++j) { mn[j] = std::min(mn[j], gx[(size_t)order[(size_t)k] * 3 + j]); mx[j] = std::max(mx[j], gx[(size_t)order[(size_t)k] * 3 + j]); }
        int axis = 0;
        for (int j = 1; j < 3; ++j)
            if (mx[j] - mn[j] > mx[axis] - mn[axis]) axis = j;
        sort_range(lo, hi, axis);        // leaves too: a patch's groups end up in lexicographic order, neighbours next to each other
        if (hi - lo <= T) continue;
        const int64_t n = hi - lo;
        int64_t left = ((n / 2 + T / 2) / T) * T;      // a multiple of T near the middle
        if (left < T) left = T;
        if (left > n - 1) left = (n - 1) / T * T;
        stack.push_back({lo + left, hi});
        stack.push_back({lo, lo + left});
    }
    // ---- incidences per node (for "is every cell of this node in the patch?")
    std::vector<int32_t> total((size_t)nn, 0);
    for (int64_t e = 0; e < nc * nd; ++e) ++total[(size_t)dofmap[(size_t)e]];
    const int64_t np = (ng + T - 1) / T;
    H.R = R;
    H.groups.assign((size_t)np * T, -1);
    H.lnode.assign((size_t)(nc * nd), 0);
    H.cellcol.assign((size_t)nc, 0);
    H.grp_ncol.assign((size_t)ng, 1);
    H.node_off.assign(1, 0);
    H.gnode.clear();
    H.mmap.clear();
    H.max_priv = 0;
    H.max_local = 0;
    std::vector<int32_t> pstamp((size_t)nn, -1), plocal((size_t)nn, 0), inpatch((size_t)nn, 0);     // patch-level numbering
    std::vector<int32_t> wstamp((size_t)nn, -1), wlocal((size_t)nn, 0);                              // wave-level numbering
    std::vector<int32_t> cstamp((size_t)nn, -1);
    std::vector<uint32_t> cmask((size_t)nn, 0);
    std::vector<std::pair<int32_t, uint32_t>> shared;    // (node, slot) of every shared patch node, in patch order
    std::vector<int32_t> pnodes;
    int64_t n_shared = 0;
    for (int64_t p = 0; p < np; ++p) {
        pnodes.clear();
        const uint32_t slot0 = (uint32_t)H.gnode.size();
        const int64_t g_lo = p * T, g_hi = std::min(ng, (p + 1) * T);
        // groups lo..hi of the order dealt to the waves in runs of R (a short patch fills the first waves)
        for (int64_t k = g_lo; k < g_hi; ++k) H.groups[(size_t)(p * T + (k - g_lo))] = order[(size_t)k];
        // pass 1: patch numbering + incidences
        for (int64_t k = g_lo; k < g_hi; ++k) {
            const int64_t g = order[(size_t)k], c1 = std::min(nc, (g + 1) * cpw);
            for (int64_t e = g * cpw * nd; e < c1 * nd; ++e) {
                const int32_t n = dofmap[(size_t)e];
                if (pstamp[(size_t)n] != (int32_t)p) { pstamp[(size_t)n] = (int32_t)p; plocal[(size_t)n] = (int32_t)pnodes.size(); inpatch[(size_t)n] = 0; pnodes.push_back(n); }
                ++inpatch[(size_t)n];
            }
        }
        H.mmap.resize(H.mmap.size() + pnodes.size() * PATCH_WAVES, (uint16_t)0xffff);
        // pass 2: wave numbering, cell colours
        for (int w = 0; w < PATCH_WAVES; ++w) {
            int32_t n_priv = 0;
            const int32_t wkey = (int32_t)(p * PATCH_WAVES + w);
            for (int r = 0; r < R; ++r) {
                const int64_t k = g_lo + (int64_t)w * R + r;
                if (k >= g_hi) break;
                const int64_t g = order[(size_t)k], c1 = std::min(nc, (g + 1) * cpw);
                int ncol = 1;
                for (int64_t c = g * cpw; c < c1; ++c) {
                    uint32_t used = 0;
                    for (int a = 0; a < nd; ++a) {
                        const int32_t n = dofmap[(size_t)(c * nd + a)];
                        if (wstamp[(size_t)n] != wkey) {
                            wstamp[(size_t)n] = wkey;
                            wlocal[(size_t)n] = n_priv;
                            H.mmap[((size_t)slot0 + (size_t)plocal[(size_t)n]) * PATCH_WAVES + (size_t)w] = (uint16_t)n_priv;
                            ++n_priv;
                        }
                        H.lnode[(size_t)(c * nd + a)] = (uint16_t)wlocal[(size_t)n];
                        if (cstamp[(size_t)n] == (int32_t)g) used |= cmask[(size_t)n];
                    }
                    int col = 0;
                    while (used & (1u << col)) ++col;
                    for (int a = 0; a < nd; ++a) {
                        const int32_t n = dofmap[(size_t)(c * nd + a)];
                        if (cstamp[(size_t)n] != (int32_t)g) { cstamp[(size_t)n] = (int32_t)g; cmask[(size_t)n] = 0; }
                        cmask[(size_t)n] |= 1u << col;
                    }
                    H.cellcol[(size_t)c] = (uint8_t)col;
                    ncol = std::max(ncol, col + 1);
                }
                H.grp_ncol[(size_t)g] = (uint8_t)ncol;
            }
            if (n_priv > priv_cap) return false;       // the caller retries with a smaller R
            H.max_priv = std::max(H.max_priv, (int)n_priv);
        }
        for (size_t ln = 0; ln < pnodes.size(); ++ln) {
            const int32_t n = pnodes[ln];
            const bool is_shared = inpatch[(size_t)n] != total[(size_t)n];
            H.gnode.push_back((uint32_t)n | (is_shared ? 0x80000000u : 0u));
            if (is_shared) { shared.emplace_back(n, slot0 + (uint32_t)ln); ++n_shared; }
        }
        H.node_off.push_back((int32_t)H.gnode.size());
        H.max_local = std::max(H.max_local, (int)pnodes.size());
        if (H.gnode.size() >= ((size_t)1 << 31)) return false;
    }
    // ---- second pass: the shared nodes' slots in patch order; untouched nodes get an empty list (they are SET to zero / left alone)
    std::vector<int32_t> cnt((size_t)nn, 0);
    for (const auto& sh : shared) ++cnt[(size_t)sh.first];
    H.bnode.clear();
    H.bptr.assign(1, 0);
    std::vector<int64_t> where((size_t)nn, -1);
    for (int64_t n = 0; n < nn; ++n)
        if (cnt[(size_t)n] > 0 || total[(size_t)n] == 0) {
            where[(size_t)n] = (int64_t)H.bnode.size();
            H.bnode.push_back((int32_t)n);
            H.bptr.push_back(H.bptr.back() + cnt[(size_t)n]);
        }
    H.bent.assign((size_t)H.bptr.back() + 3, 0u);
    std::vector<int64_t> fill(H.bptr.begin(), H.bptr.end() - 1);
    for (const auto& sh : shared) H.bent[(size_t)fill[(size_t)where[(size_t)sh.first]]++] = sh.second;
    H.shared_fraction = (double)n_shared / (double)(nc * nd);
    return true;
}

// second pass of the patch form: out[node] (+)= the node's partials, in patch order
template <int BS>
__global__ __launch_bounds__(DXO_BLOCK) void node_sum_patch(PatchDev P, double* __restrict__ out, int overwrite) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < P.n_bnodes; k += stride) {
        const int64_t n = P.bnode[k], e0 = P.bptr[k], e1 = P.bptr[k + 1];
        double acc[BS], cur[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            acc[i] = 0.0;
            cur[i] = overwrite ? 0.0 : out[n * BS + i];
        }
        for (int64_t e = e0; e < e1; e += 4) {
            uint32_t idx[4];
            double f[4][BS];
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = e + j < e1 ? P.bent[e + j] : P.bent[e0];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < BS; ++i) f[j][i] = P.bpart[(int64_t)idx[j] * BS + i];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (e + j < e1) {
#pragma unroll
                    for (int i = 0; i < BS; ++i) acc[i] += f[j][i];
                }
        }
#pragma unroll
        for (int i = 0; i < BS; ++i) out[n * BS + i] = cur[i] + acc[i];
    }
}

// ---- device side of a patch: one accumulator per wave in LDS, merged at the end of the patch
template <int BS>
struct PatchAcc {
    double* base;         // [PATCH_WAVES][max_priv][BS]
    int max_priv;
    __device__ __forceinline__ double* mine(int wave) const { return base + (size_t)wave * max_priv * BS; }
    // the wave clears its own accumulator (after the barrier that ends the previous patch's merge)
    __device__ __forceinline__ void zero(int wave, int lane) const {
        double* a = mine(wave);
        for (int k = lane; k < max_priv * BS; k += DXO_WAVE) a[k] = 0.0;
    }
    // one read-add-write pass per cell colour: lanes whose cell has colour `col` add `o` to entry `ln` of the wave's accumulator.
    // All lanes of the wave call this (uniform control flow); `has` masks the lanes that have something to add. LDS operations of
    // one wave execute in program order: a pass sees the sums of the one before.
    __device__ __forceinline__ void add(double* a, int ncol, int col_l, bool has, int ln, const double (&o)[BS]) const {
        for (int col = 0; col < ncol; ++col) {
            if (has && col_l == col) {
#pragma unroll
                for (int i = 0; i < BS; ++i) a[ln * BS + i] += o[i];
            }
        }
    }
    // after the barrier that ends the patch's accumulation: per patch node the waves' partials in wave order; unshared nodes go to
    // the dof vector, shared ones leave the patch's partial
    __device__ __forceinline__ void merge_flush(const PatchDev& P, int patch, double* __restrict__ out, int overwrite) const {
        const int off = P.node_off[patch], n_local = P.node_off[patch + 1] - off;
        for (int ln = threadIdx.x; ln < n_local; ln += blockDim.x) {
            const uint32_t g = P.gnode[off + ln];
            const uint16_t* mm = P.mmap + (size_t)(off + ln) * PATCH_WAVES;
            double v[BS];
#pragma unroll
            for (int i = 0; i < BS; ++i) v[i] = 0.0;
#pragma unroll
            for (int w = 0; w < PATCH_WAVES; ++w) {
                const int k = mm[w];
                if (k != 0xffff) {
                    const double* a = mine(w) + k * BS;
#pragma unroll
                    for (int i = 0; i < BS; ++i) v[i] += a[i];
                }
            }
            if (g & 0x80000000u) {
#pragma unroll
                for (int i = 0; i < BS; ++i) P.bpart[(int64_t)(off + ln) * BS + i] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < BS; ++i) {
                    const int64_t k = (int64_t)g * BS + i;
                    out[k] = overwrite ? v[i] : out[k] + v[i];
                }
            }
        }
    }
};

}  // namespace
