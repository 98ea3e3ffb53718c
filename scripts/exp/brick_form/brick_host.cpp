// Host half of the brick form (csrc/cell8_brick_host.h) on its own: reads a mesh from stdin, builds the bricks, replays the wave's
// reduction + node_sum on the CPU for random element vectors and compares with the plain per-cell sum. Built and run by tests/test_brick_host.py.
#include <cstdio>
#include <random>

#include "cell8_brick_host.h"

int main() {
    long long nc, nd, nn;
    if (std::scanf("%lld %lld %lld", &nc, &nd, &nn) != 3) return 2;
    std::vector<int32_t> dofmap((size_t)(nc * nd)), geom((size_t)(nc * 8));
    std::vector<float> xyz((size_t)(nc * 3));
    for (auto& v : dofmap) if (std::scanf("%d", &v) != 1) return 2;
    for (auto& v : geom) if (std::scanf("%d", &v) != 1) return 2;
    for (auto& v : xyz) if (std::scanf("%f", &v) != 1) return 2;
    double ext[3];
    if (std::scanf("%lf %lf %lf", &ext[0], &ext[1], &ext[2]) != 3) return 2;
    BrickHost H;
    if (!brick_build_host(dofmap, geom, xyz, ext, nc, (int)nd, nn, H)) { std::printf("{\"ok\": false}\n"); return 0; }
    const long long ng = (nc + 7) / 8;
    // element vectors fe[cell][a][i] in the CALLER's cell order
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> dist(-1.0, 1.0);
    std::vector<double> fe((size_t)(nc * nd * 3));
    for (auto& v : fe) v = dist(rng);
    std::vector<double> ref((size_t)(nn * 3), 0.0), got((size_t)(nn * 3), 0.0), part((size_t)(H.n_slots * 3), 0.0);
    for (long long c = 0; c < nc; ++c)
        for (int a = 0; a < nd; ++a)
            for (int i = 0; i < 3; ++i) ref[(size_t)dofmap[(size_t)(c * nd + a)] * 3 + i] += fe[(size_t)((c * nd + a) * 3 + i)];
    long long min_u = 1 << 30, max_u = 0, bad = 0;
    for (long long g = 0; g < ng; ++g) {
        const uint8_t* tab = H.tabs.data() + (size_t)H.tab_id[(size_t)g] * BRICK_TAB_BYTES_H;
        const int U = tab[0];
        min_u = std::min<long long>(min_u, U);
        max_u = std::max<long long>(max_u, U);
        const int ncell = (int)std::min<long long>(8, nc - g * 8);
        // the staged result D[a][3 c + i] of the group (brick order position g * 8 + c -> caller's cell orig[])
        std::vector<double> D((size_t)(27 * BRICK_DS + 8), 0.0);
        for (int c = 0; c < ncell; ++c) {
            const long long oc = H.orig[(size_t)(g * 8 + c)];
            for (int a = 0; a < nd; ++a) {
                if (H.dofmap[(size_t)((g * 8 + c) * nd + a)] != dofmap[(size_t)(oc * nd + a)]) ++bad;
                for (int i = 0; i < 3; ++i) D[(size_t)(a * BRICK_DS + 3 * c + i)] = fe[(size_t)((oc * nd + a) * 3 + i)];
            }
        }
        const uint16_t* src = reinterpret_cast<const uint16_t*>(tab + BRICK_TAB_SRC_H);
        for (int u = 0; u < U; ++u)
            for (int i = 0; i < 3; ++i) {
                double sum = 0.0;
                for (int s = tab[BRICK_TAB_START_H + u]; s < tab[BRICK_TAB_START_H + u + 1]; ++s) sum += D[(size_t)(src[s] + i)];
                part[(size_t)((H.slot0[(size_t)g] + u) * 3 + i)] = sum;
            }
        if (tab[BRICK_TAB_START_H + U] != ncell * nd) ++bad;      // every entry of the group is in exactly one slot
    }
    for (long long n = 0; n < nn; ++n)
        for (long long e = H.node_ptr[(size_t)n]; e < H.node_ptr[(size_t)n + 1]; ++e)
            for (int i = 0; i < 3; ++i) got[(size_t)(n * 3 + i)] += part[(size_t)H.node_ent[(size_t)e] * 3 + i];
    double err = 0.0;
    for (size_t k = 0; k < ref.size(); ++k) err = std::max(err, std::fabs(ref[k] - got[k]));
    long long max_part = 0;
    for (long long n = 0; n < nn; ++n) max_part = std::max<long long>(max_part, H.node_ptr[(size_t)n + 1] - H.node_ptr[(size_t)n]);
    std::printf("{\"ok\": true, \"groups\": %lld, \"slots\": %lld, \"tables\": %lld, \"min_u\": %lld, \"max_u\": %lld, \"max_partials_per_node\": %lld, \"bad\": %lld, \"err\": %.3e}\n",
                ng, (long long)H.n_slots, (long long)(H.tabs.size() / BRICK_TAB_BYTES_H), min_u, max_u, max_part, bad, err);
    return 0;
}
