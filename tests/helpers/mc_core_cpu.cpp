// Host build of csrc/mc_core.h — TEST AID ONLY: lets the per-lane Mohr-Coulomb math be checked against the
// oracle and the goldens in the CPU-only container. Never loaded by the product package.
#include <cstdint>
#include <cstring>

#include "mc_core.h"

struct Prm { double E, nu, c, phi, psi, theta_T, a, tol; int32_t nitermax, pad; };

extern "C" int mc_core_cpu(const void* prm, int64_t n, const double* deps, const double* sigma_n, double* C_tang,
                           double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda) {
    Prm p;
    std::memcpy(&p, prm, sizeof p);
    const mc::Const k = mc::make_const(p.E, p.nu, p.c, p.phi, p.psi, p.theta_T, p.a, p.tol, p.nitermax);
    for (int64_t i = 0; i < n; ++i) {
        mc::Result R;
        mc::return_map(k, deps + 4 * i, sigma_n + 4 * i, R);
        std::memcpy(C_tang + 16 * i, R.C_tang, sizeof R.C_tang);
        std::memcpy(sigma + 4 * i, R.sigma, sizeof R.sigma);
        niter[i] = R.niter;
        yielding[i] = R.yielding;
        norm_res[i] = R.norm_res;
        dlambda[i] = R.dlambda;
    }
    return 0;
}
