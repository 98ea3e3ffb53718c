// cell8_brick_host.h — host half of the brick form (cell8_brick.h): cells -> Morton order -> wave groups of 8 -> reduction tables and the
// transposed map node -> group partials. Plain C++17, no HIP types: also compiled on its own by tests/test_brick_host.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

// layout of one reduction table (BRICK_TAB_BYTES): [0] U = slots of the group; [4 + u] start[u], u = 0 .. U (the entries of slot u are
// start[u] .. start[u + 1]); [224 + 2 s] uint16 position of entry s in the staged result D: a * BRICK_DS + 3 c (the lane adds its component)
constexpr int BRICK_TAB_BYTES_H = 704, BRICK_TAB_START_H = 4, BRICK_TAB_SRC_H = 224;      // = BRICK_TAB_* of operand_core.h (asserted in cell8_brick.h)
constexpr int BRICK_DS = 25;       // row stride of the staged MFMA result D[node a][(cell, component)] in the wave's LDS slice (doubles)

namespace {

inline uint64_t brick_spread3(uint64_t v) {      // 21 bits -> every third bit
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

struct BrickHost {
    std::vector<int32_t> orig, dofmap, geom, tab_id;
    std::vector<int64_t> slot0, node_ptr;
    std::vector<uint32_t> node_ent;
    std::vector<uint8_t> tabs;
    int64_t n_slots = 0;
};

// cells -> Morton order of their centroids -> groups of 8 -> per group: unique nodes (slots, in order of first appearance), the
// reduction table (per slot the LDS positions of its entries, ascending cell), and the transposed map node -> slots
inline bool brick_build_host(const std::vector<int32_t>& dofmap, const std::vector<int32_t>& geom, const std::vector<float>& xyz, const double (&ext)[3],
                             int64_t nc, int nd, int64_t n_nodes, BrickHost& H) {
    if (nc <= 0 || (nd != 27 && nd != 8) || (int64_t)dofmap.size() != nc * nd || (int64_t)geom.size() != nc * 8 || (int64_t)xyz.size() != nc * 3) return false;
    float lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (int64_t c = 0; c < nc; ++c)
        for (int j = 0; j < 3; ++j) {
            lo[j] = std::min(lo[j], xyz[(size_t)c * 3 + j]);
            hi[j] = std::max(hi[j], xyz[(size_t)c * 3 + j]);
        }
    // grid spacing per axis: the mean extent of a cell along that axis (`ext`, from the mesh's vertices) — on a structured mesh exactly
    // the spacing, so that the centroids sit at (i + 1/2) h and the low bit of every axis index pairs the cells of a brick
    double h[3];
    for (int j = 0; j < 3; ++j) h[j] = ext[j] > 0.0 ? ext[j] : std::max(((double)hi[j] - lo[j]) / std::cbrt((double)nc), 1e-30);
    std::vector<std::pair<uint64_t, int32_t>> key((size_t)nc);
    for (int64_t c = 0; c < nc; ++c) {
        uint64_t code = 0;
        for (int j = 0; j < 3; ++j) {
            double t = ((double)xyz[(size_t)c * 3 + j] - lo[j]) / h[j] + 0.25;      // centroids sit near 0, 1, 2, ... in units of h
            if (t < 0) t = 0;
            code |= brick_spread3((uint64_t)t) << j;
        }
        key[(size_t)c] = {code, (int32_t)c};
    }
    std::sort(key.begin(), key.end());
    H.orig.resize((size_t)nc);
    H.dofmap.resize((size_t)nc * nd);
    H.geom.resize((size_t)nc * 8);
    for (int64_t kx = 0; kx < nc; ++kx) {
        const int32_t c = key[(size_t)kx].second;
        H.orig[(size_t)kx] = c;
        std::memcpy(&H.dofmap[(size_t)kx * nd], &dofmap[(size_t)c * nd], (size_t)nd * sizeof(int32_t));
        std::memcpy(&H.geom[(size_t)kx * 8], &geom[(size_t)c * 8], 8 * sizeof(int32_t));
    }
    const int64_t ng = (nc + 7) / 8;
    H.tab_id.resize((size_t)ng);
    H.slot0.resize((size_t)ng);
    std::vector<int64_t> stamp((size_t)n_nodes, -1);
    std::vector<int32_t> slot_of((size_t)n_nodes, 0);
    std::unordered_map<std::string, int32_t> seen;
    std::vector<int64_t> count((size_t)n_nodes + 1, 0);
    std::vector<int32_t> grp_nodes;              // per group: the node of every slot, concatenated (for the transposed map)
    grp_nodes.reserve((size_t)ng * 125);
    int64_t slots = 0;
    uint8_t tab[BRICK_TAB_BYTES_H];
    for (int64_t g = 0; g < ng; ++g) {
        const int ncell = (int)std::min<int64_t>(8, nc - g * 8);
        int U = 0;
        int cnt[216];
        int16_t ent_slot[216];
        for (int c = 0; c < ncell; ++c)
            for (int a = 0; a < nd; ++a) {
                const int32_t node = H.dofmap[(size_t)(g * 8 + c) * nd + a];
                if (stamp[(size_t)node] != g) {
                    stamp[(size_t)node] = g;
                    slot_of[(size_t)node] = U;
                    cnt[U] = 0;
                    grp_nodes.push_back(node);
                    ++count[(size_t)node + 1];
                    ++U;
                }
                ent_slot[c * nd + a] = (int16_t)slot_of[(size_t)node];
                ++cnt[slot_of[(size_t)node]];
            }
        std::memset(tab, 0, sizeof tab);
        tab[0] = (uint8_t)U;
        int start[217];
        start[0] = 0;
        for (int u = 0; u < U; ++u) start[u + 1] = start[u] + cnt[u];
        for (int u = 0; u <= U; ++u) tab[BRICK_TAB_START_H + u] = (uint8_t)start[u];
        int fill[216];
        for (int u = 0; u < U; ++u) fill[u] = start[u];
        uint16_t* src = reinterpret_cast<uint16_t*>(tab + BRICK_TAB_SRC_H);
        for (int c = 0; c < ncell; ++c)           // ascending cell: the fixed order of every slot's sum
            for (int a = 0; a < nd; ++a) src[fill[ent_slot[c * nd + a]]++] = (uint16_t)(a * BRICK_DS + c * 3);
        const std::string keyb(reinterpret_cast<const char*>(tab), sizeof tab);
        auto it = seen.find(keyb);
        if (it == seen.end()) {
            it = seen.emplace(keyb, (int32_t)seen.size()).first;
            H.tabs.insert(H.tabs.end(), tab, tab + sizeof tab);
        }
        H.tab_id[(size_t)g] = it->second;
        H.slot0[(size_t)g] = slots;
        slots += U;
    }
    if (slots >= ((int64_t)1 << 32)) return false;       // uint32 slot ids in the transposed map
    H.n_slots = slots;
    // transposed map: node -> slot ids of its partials, ascending group order
    H.node_ptr.assign(count.begin(), count.end());
    for (int64_t n = 0; n < n_nodes; ++n) H.node_ptr[(size_t)n + 1] += H.node_ptr[(size_t)n];
    H.node_ent.assign((size_t)slots, 0);
    std::vector<int64_t> fillp(H.node_ptr.begin(), H.node_ptr.end() - 1);
    for (int64_t s = 0; s < slots; ++s) H.node_ent[(size_t)fillp[(size_t)grp_nodes[(size_t)s]]++] = (uint32_t)s;
    return true;
}

}  // namespace
