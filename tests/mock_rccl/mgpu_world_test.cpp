// mgpu_world_test.cpp — the N > 1 data path of libdxo (dxo_mgpu_von_mises, csrc/mgpu.hip) on ONE GPU over the mock transport
// (mock_rccl.cpp): `world` ranks of one process, all on device 0. For every gather form the full-length arrays of EVERY rank must
// equal, bit for bit, the concatenation of the blocks computed one at a time by a world-of-one group in the same form.
// usage: mgpu_world_test <world> <n_per_rank>; prints one "ok ..." line per form, exit code 0 when all match.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dxo.h"

extern "C" long mock_rccl_bytes_moved();
enum { DXO_COPY_H2D = 0, DXO_COPY_D2H = 1 };      // dxo_copy kinds (include/dxo.h)

#define CHECK(call)                                                                                           \
    do {                                                                                                      \
        int rc_ = (call);                                                                                     \
        if (rc_ != 0) { std::fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, g ? dxo_mgpu_last_error(g) : ""); return 1; } \
    } while (0)

static double lcg(uint64_t& s) {      // uniform in (-1, 1)
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 52) - 1.0;
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 3;
    const int64_t n = argc > 2 ? std::atoll(argv[2]) : 10000;
    const int d = 6;
    dxo_mgpu* g = nullptr;
    const double E = 70e3, Et = E / 100.0;
    const dxo_vm_params prm = {E, 0.3, 250.0, E * Et / (E - Et)};
    std::vector<int> devs((size_t)world, 0);
    setenv("DXO_MGPU_TEST_SHARE_DEVICE", "1", 1);
    CHECK(dxo_mgpu_create(devs.data(), world, &g));
    dxo_mgpu* one = nullptr;
    { const int z = 0; int rc = dxo_mgpu_create(&z, 1, &one); if (rc) { std::fprintf(stderr, "world-of-one group: %d\n", rc); return 1; } }
    // inputs: a mix of elastic and plastic points, different on every rank
    std::vector<std::vector<double>> deps(world), sig(world), pp(world);
    uint64_t seed = 12345;
    for (int r = 0; r < world; ++r) {
        deps[r].resize(n * d); sig[r].resize(n * d); pp[r].resize(n);
        for (int64_t i = 0; i < n * d; ++i) { deps[r][i] = 3e-3 * lcg(seed); sig[r][i] = 100.0 * lcg(seed); }
        for (int64_t i = 0; i < n; ++i) pp[r][i] = std::fabs(1e-3 * lcg(seed));
    }
    auto dev_alloc = [&](dxo_ctx* c, size_t bytes, const void* host) -> double* {
        void* p = nullptr;
        if (dxo_device_alloc(c, (int64_t)bytes, &p) != 0) return nullptr;
        if (host && dxo_copy(c, p, host, (int64_t)bytes, DXO_COPY_H2D) != 0) return nullptr;
        return static_cast<double*>(p);
    };
    std::vector<double*> d_deps(world), d_sig(world), d_p(world), d_C(world), d_s(world), d_dp(world);
    const size_t NC = (size_t)world * n * d * d, NS = (size_t)world * n * d, NP = (size_t)world * n;
    for (int r = 0; r < world; ++r) {
        dxo_ctx* c = dxo_mgpu_ctx(g, r);
        d_deps[r] = dev_alloc(c, n * d * 8, deps[r].data()); d_sig[r] = dev_alloc(c, n * d * 8, sig[r].data()); d_p[r] = dev_alloc(c, n * 8, pp[r].data());
        d_C[r] = dev_alloc(c, NC * 8, nullptr); d_s[r] = dev_alloc(c, NS * 8, nullptr); d_dp[r] = dev_alloc(c, NP * 8, nullptr);
        if (!d_deps[r] || !d_sig[r] || !d_p[r] || !d_C[r] || !d_s[r] || !d_dp[r]) { std::fprintf(stderr, "device allocation failed\n"); return 1; }
    }
    dxo_ctx* c1 = dxo_mgpu_ctx(one, 0);
    double *o_deps = dev_alloc(c1, n * d * 8, nullptr), *o_sig = dev_alloc(c1, n * d * 8, nullptr), *o_p = dev_alloc(c1, n * 8, nullptr);
    double *o_C = dev_alloc(c1, (size_t)n * d * d * 8, nullptr), *o_s = dev_alloc(c1, n * d * 8, nullptr), *o_dp = dev_alloc(c1, n * 8, nullptr);
    struct Form { int gather; int chunks; const char* name; };
    const Form forms[] = {{DXO_GATHER_FULL, 4, "full"}, {DXO_GATHER_COMPACT, 4, "compact"}, {DXO_GATHER_COMPACT_DIRECT, 4, "compact_direct"},
                          {DXO_GATHER_COMPACT_PIPELINED, 1, "compact_pipelined/1"}, {DXO_GATHER_COMPACT_PIPELINED, 4, "compact_pipelined/4"},
                          {DXO_GATHER_COMPACT_PIPELINED, 7, "compact_pipelined/7"}};
    int bad = 0;
    std::vector<double> refC(NC), refS(NS), refP(NP), gotC(NC), gotS(NS), gotP(NP);
    const double poison = -12345.678;
    for (const Form& f : forms) {
        // reference: every block through a world-of-one group in the same form
        for (int r = 0; r < world; ++r) {
            if (dxo_copy(c1, o_deps, deps[r].data(), n * d * 8, DXO_COPY_H2D) || dxo_copy(c1, o_sig, sig[r].data(), n * d * 8, DXO_COPY_H2D) ||
                dxo_copy(c1, o_p, pp[r].data(), n * 8, DXO_COPY_H2D)) return 1;
            const double* a[1] = {o_deps}; const double* b[1] = {o_sig}; const double* c[1] = {o_p};
            double* C[1] = {o_C}; double* s[1] = {o_s}; double* q[1] = {o_dp};
            (void)dxo_ctx_set_option(c1, "mgpu_chunks", f.chunks);
            int rc = dxo_mgpu_von_mises(one, &prm, d, n, f.gather, a, b, c, C, s, q);
            if (rc) { std::fprintf(stderr, "reference %s: %d %s\n", f.name, rc, dxo_mgpu_last_error(one)); return 1; }
            if (dxo_mgpu_synchronize(one)) return 1;
            if (dxo_copy(c1, refC.data() + (size_t)r * n * d * d, o_C, (int64_t)n * d * d * 8, DXO_COPY_D2H) ||
                dxo_copy(c1, refS.data() + (size_t)r * n * d, o_s, n * d * 8, DXO_COPY_D2H) || dxo_copy(c1, refP.data() + (size_t)r * n, o_dp, n * 8, DXO_COPY_D2H)) return 1;
        }
        // the group: poison the outputs first
        std::vector<double> fill(NC, poison);
        for (int r = 0; r < world; ++r) {
            dxo_ctx* c = dxo_mgpu_ctx(g, r);
            if (dxo_copy(c, d_C[r], fill.data(), NC * 8, DXO_COPY_H2D) || dxo_copy(c, d_s[r], fill.data(), NS * 8, DXO_COPY_H2D) || dxo_copy(c, d_dp[r], fill.data(), NP * 8, DXO_COPY_H2D)) return 1;
            (void)dxo_ctx_set_option(c, "mgpu_chunks", f.chunks);
        }
        const long before = mock_rccl_bytes_moved();
        std::vector<const double*> a(d_deps.begin(), d_deps.end()), b(d_sig.begin(), d_sig.end()), c(d_p.begin(), d_p.end());
        CHECK(dxo_mgpu_von_mises(g, &prm, d, n, f.gather, a.data(), b.data(), c.data(), d_C.data(), d_s.data(), d_dp.data()));
        CHECK(dxo_mgpu_synchronize(g));
        const long moved = mock_rccl_bytes_moved() - before;
        const long per_pt = f.gather == DXO_GATHER_FULL ? 8 * (d * d + d + 1) : 8 * (d + 1);
        const long expect = (long)world * (world - 1) * n * per_pt;
        int form_bad = moved != expect;
        if (form_bad) std::fprintf(stderr, "%s: %ld bytes over the transport, expected %ld\n", f.name, moved, expect);
        for (int r = 0; r < world; ++r) {
            dxo_ctx* cx = dxo_mgpu_ctx(g, r);
            if (dxo_copy(cx, gotC.data(), d_C[r], NC * 8, DXO_COPY_D2H) || dxo_copy(cx, gotS.data(), d_s[r], NS * 8, DXO_COPY_D2H) || dxo_copy(cx, gotP.data(), d_dp[r], NP * 8, DXO_COPY_D2H)) return 1;
            if (std::memcmp(gotC.data(), refC.data(), NC * 8) || std::memcmp(gotS.data(), refS.data(), NS * 8) || std::memcmp(gotP.data(), refP.data(), NP * 8)) {
                size_t k = 0;
                while (k < NS && !std::memcmp(&gotS[k], &refS[k], 8)) ++k;
                std::fprintf(stderr, "%s: rank %d differs from the block-by-block reference (first sigma difference at element %zu of %zu)\n", f.name, r, k, NS);
                form_bad = 1;
            }
        }
        std::printf("%s %s world %d n_per_rank %lld: %ld bytes moved\n", form_bad ? "MISMATCH" : "ok", f.name, world, (long long)n, moved);
        bad += form_bad;
    }
    // ---- another operator through the same split + all-gather: Mohr-Coulomb, five output arrays of three element sizes (16, 4
    // doubles, an int32 and two doubles per point), one of them not requested (NULL pointer array)
    {
        const double phi = 0.5235987755982988;
        const dxo_mc_params mc = {6778.0, 0.25, 3.45, phi, phi, 26.0 * 3.141592653589793 / 180.0, 0.26 * 3.45 / std::tan(phi), 1e-8, 200, 0};
        const int64_t m = n - (n % 4);
        std::vector<std::vector<double>> e(world), s0(world);
        for (int r = 0; r < world; ++r) {
            e[r].resize(m * 4); s0[r].resize(m * 4);
            for (int64_t i = 0; i < m; ++i) {
                const double load = (i % 3 == 0) ? 2e-3 : 1e-5;         // a third of the points yield
                for (int k = 0; k < 4; ++k) { e[r][i * 4 + k] = load * lcg(seed); s0[r][i * 4 + k] = (k < 3 ? -1.0 : 0.1) * std::fabs(lcg(seed)); }
            }
        }
        std::vector<double*> me(world), ms(world), mC(world), mS(world), mY(world), mL(world);
        std::vector<int32_t*> mI(world);
        const size_t M = (size_t)world * m;
        for (int r = 0; r < world; ++r) {
            dxo_ctx* c = dxo_mgpu_ctx(g, r);
            me[r] = dev_alloc(c, m * 32, e[r].data()); ms[r] = dev_alloc(c, m * 32, s0[r].data());
            mC[r] = dev_alloc(c, M * 128, nullptr); mS[r] = dev_alloc(c, M * 32, nullptr); mY[r] = dev_alloc(c, M * 8, nullptr); mL[r] = dev_alloc(c, M * 8, nullptr);
            mI[r] = reinterpret_cast<int32_t*>(dev_alloc(c, M * 4, nullptr));
        }
        double *oe = dev_alloc(c1, m * 32, nullptr), *os = dev_alloc(c1, m * 32, nullptr), *oC = dev_alloc(c1, m * 128, nullptr), *oS = dev_alloc(c1, m * 32, nullptr);
        double *oY = dev_alloc(c1, m * 8, nullptr), *oL = dev_alloc(c1, m * 8, nullptr);
        int32_t* oI = reinterpret_cast<int32_t*>(dev_alloc(c1, m * 4, nullptr));
        std::vector<double> rC(M * 16), rS(M * 4), rY(M), rL(M), gC(M * 16), gS(M * 4), gY(M), gL(M);
        std::vector<int32_t> rI(M), gI(M);
        for (int r = 0; r < world; ++r) {
            if (dxo_copy(c1, oe, e[r].data(), m * 32, DXO_COPY_H2D) || dxo_copy(c1, os, s0[r].data(), m * 32, DXO_COPY_H2D)) return 1;
            int rc = dxo_mohr_coulomb(c1, &mc, m, DXO_MEM_DEVICE, oe, os, oC, oS, oI, oY, nullptr, oL);
            if (rc || dxo_ctx_synchronize(c1)) { std::fprintf(stderr, "Mohr-Coulomb reference: %d\n", rc); return 1; }
            if (dxo_copy(c1, rC.data() + (size_t)r * m * 16, oC, m * 128, DXO_COPY_D2H) || dxo_copy(c1, rS.data() + (size_t)r * m * 4, oS, m * 32, DXO_COPY_D2H) ||
                dxo_copy(c1, rI.data() + (size_t)r * m, oI, m * 4, DXO_COPY_D2H) || dxo_copy(c1, rY.data() + (size_t)r * m, oY, m * 8, DXO_COPY_D2H) ||
                dxo_copy(c1, rL.data() + (size_t)r * m, oL, m * 8, DXO_COPY_D2H)) return 1;
        }
        std::vector<const double*> a(me.begin(), me.end()), b(ms.begin(), ms.end());
        CHECK(dxo_mgpu_mohr_coulomb(g, &mc, m, DXO_GATHER_FULL, a.data(), b.data(), mC.data(), mS.data(), mI.data(), mY.data(), nullptr, mL.data()));
        CHECK(dxo_mgpu_synchronize(g));
        int mc_bad = 0;
        long plastic = 0;
        for (size_t i = 0; i < M; ++i) plastic += rY[i] > 0.0;
        for (int r = 0; r < world; ++r) {
            dxo_ctx* cx = dxo_mgpu_ctx(g, r);
            if (dxo_copy(cx, gC.data(), mC[r], M * 128, DXO_COPY_D2H) || dxo_copy(cx, gS.data(), mS[r], M * 32, DXO_COPY_D2H) || dxo_copy(cx, gI.data(), mI[r], M * 4, DXO_COPY_D2H) ||
                dxo_copy(cx, gY.data(), mY[r], M * 8, DXO_COPY_D2H) || dxo_copy(cx, gL.data(), mL[r], M * 8, DXO_COPY_D2H)) return 1;
            if (std::memcmp(gC.data(), rC.data(), M * 128) || std::memcmp(gS.data(), rS.data(), M * 32) || std::memcmp(gI.data(), rI.data(), M * 4) ||
                std::memcmp(gY.data(), rY.data(), M * 8) || std::memcmp(gL.data(), rL.data(), M * 8)) {
                std::fprintf(stderr, "mohr_coulomb full gather: rank %d differs from the block-by-block reference\n", r);
                mc_bad = 1;
            }
        }
        std::printf("%s mohr_coulomb_full world %d n_per_rank %lld: %ld of %zu points plastic\n", mc_bad ? "MISMATCH" : "ok", world, (long long)m, plastic, M);
        bad += mc_bad;
    }
    dxo_mgpu_destroy(g);
    dxo_mgpu_destroy(one);
    return bad ? 1 : 0;
}
