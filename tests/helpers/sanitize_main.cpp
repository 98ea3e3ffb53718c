// Sanitizer harness (CPU only): runs the CPU oracle and the host build of the Mohr-Coulomb lane math on a few
// thousand seeded points under -fsanitize=address,undefined. GPU AddressSanitizer is not available on the pool,
// so memory-safety / UB checking of the shared per-point math happens here.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mc_core.h"

extern "C" {
int oracle_von_mises(const double*, int, int64_t, const double*, const double*, const double*, double*, double*, double*, int);
int oracle_heat(double, double, int, int64_t, const double*, const double*, double*, double*, double*, int);
int oracle_mohr_coulomb(const void*, int64_t, const double*, const double*, double*, double*, int32_t*, double*, double*, double*, int);
}

struct Prm { double E, nu, c, phi, psi, theta_T, a, tol; int32_t nitermax, pad; };

static double rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) / 16777216.0) * 2.0 - 1.0; }

int main() {
    unsigned seed = 12345u;
    // von Mises d = 4 and 6, sizes that are not multiples of anything
    for (int d : {4, 6}) {
        const int64_t n = 1237;
        std::vector<double> e(n * d), s(n * d), p(n), C(n * d * d), sg(n * d), dp(n);
        for (auto& v : e) v = rnd(seed) * 5e-3;
        for (auto& v : s) v = rnd(seed) * 150.0;
        for (auto& v : p) v = std::fabs(rnd(seed)) * 1e-3;
        const double prm[4] = {70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0)};
        if (oracle_von_mises(prm, d, n, e.data(), s.data(), p.data(), C.data(), sg.data(), dp.data(), 2)) return 1;
    }
    {
        const int64_t n = 777;
        for (int g = 1; g <= 3; ++g) {
            std::vector<double> T(n), s(n * g), q(n * g), dT(n * g), ds(n * g * g);
            for (auto& v : T) v = 1.0 + rnd(seed) * 0.5;
            for (auto& v : s) v = rnd(seed);
            if (oracle_heat(1.0, 1.0, g, n, T.data(), s.data(), q.data(), dT.data(), ds.data(), 1)) return 2;
        }
    }
    {
        const int64_t n = 400;
        const double phi = M_PI / 6;
        Prm prm{6778.0, 0.25, 3.45, phi, phi, 26 * M_PI / 180, 0.26 * 3.45 / std::tan(phi), 1e-8, 50, 0};
        std::vector<double> e(n * 4), s(n * 4), C(n * 16), sg(n * 4), y(n), nr(n), dl(n);
        std::vector<int32_t> it(n);
        for (int64_t i = 0; i < n; ++i) {
            const double pb = -2.0 + rnd(seed);
            for (int k = 0; k < 3; ++k) s[i * 4 + k] = pb + 0.5 * rnd(seed);
            s[i * 4 + 3] = 0.2 * rnd(seed);
            for (int k = 0; k < 4; ++k) e[i * 4 + k] = 2e-4 * rnd(seed);
        }
        if (oracle_mohr_coulomb(&prm, n, e.data(), s.data(), C.data(), sg.data(), it.data(), y.data(), nr.data(), dl.data(), 1)) return 3;
        const mc::Const k = mc::make_const(prm.E, prm.nu, prm.c, prm.phi, prm.psi, prm.theta_T, prm.a, prm.tol, prm.nitermax);
        long mismatched = 0;
        for (int64_t i = 0; i < n; ++i) {
            mc::Result R;
            mc::return_map(k, e.data() + 4 * i, s.data() + 4 * i, R);
            if (R.niter != it[i]) ++mismatched;
        }
        std::printf("sanitize harness: mc niter mismatches %ld of %ld\n", mismatched, (long)n);
        if (mismatched) return 4;
    }
    std::puts("sanitize harness: ok");
    return 0;
}
