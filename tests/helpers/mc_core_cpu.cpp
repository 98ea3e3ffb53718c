// Host build of csrc/mc_core.h — TEST AID ONLY: lets the per-lane Mohr-Coulomb math be checked against the
// oracle and the goldens in the CPU-only container. Never loaded by the product package.
#include <cstdint>
#include <cmath>
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

// The pass applies hess(g) and T(t) = D_t hess(g) as dense symmetric matrices (hess_dense, third_dense); the structured
// operators they replaced (hess_apply, third_setup / third_apply: literal chain-rule forms) stay in mc_core.h as their
// cross-check. out[0..3] = H v dense, [4..7] = H v structured, [8..11] = T(t) v dense, [12..15] = T(t) v structured.
extern "C" int mc_dense_vs_structured(const void* prm, const double* sig, const double* t, const double* v, double* out) {
    Prm p;
    std::memcpy(&p, prm, sizeof p);
    const mc::Const k = mc::make_const(p.E, p.nu, p.c, p.phi, p.psi, p.theta_T, p.a, p.tol, p.nitermax);
    mc::Surf e;
    mc::surf_eval<false>(k, sig, e);
    mc::Sym4 H, T;
    double a[4], b[4];
    mc::hess_dense(e, 1, H, a, b);
    mc::sym_apply(H, v, out);
    mc::hess_apply(e, 1, v, out + 4);
    mc::third_dense(e, 1, t, a, b, T);
    mc::sym_apply(T, v, out + 8);
    mc::Third S;
    mc::third_setup(e, 1, t, S);
    mc::third_apply(e, 1, S, v, out + 12);
    return 0;
}

// sin(asin(u) / 3), cos(asin(u) / 3) as the pass computes them (roots of the triple-angle cubics)
extern "C" int mc_lode_sin_cos(int64_t n, const double* u, double* sn, double* cs) {
    for (int64_t i = 0; i < n; ++i) mc::lode_sin_cos(u[i], std::sqrt((1.0 - u[i]) * (1.0 + u[i])), sn[i], cs[i]);
    return 0;
}
