/*
 * dxo_oracle.c — CPU restatement of the reference's per-quadrature-point kernels.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the parity oracle: it may be compiled, linked, imported
 * or executed only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. The product
 * path (dolfinx_external_operator_amd + libdxo_hip.so) never calls it and has no CPU fallback.
 *
 * Parity status
 *   von Mises : PINNED — checked against tests/golden/von_mises_d{4,6}.npz, which were produced by
 *               executing the reference's own `return_mapping` / `_kernel`
 *               (doc/demo/demo_plasticity_von_mises.py:298-332) in the build container
 *               (tests/golden/make_golden_von_mises.py).
 *   heat      : PINNED — checked against tests/golden/heat_c1.npz produced by executing the reference's
 *               q_impl / dqdT_impl / dqdsigma_impl (doc/demo/demo_nonlinear_heat_equation_part2.py:215-261)
 *               (tests/golden/make_golden_heat.py).
 *   conductivity : PINNED — checked against tests/golden/conductivity_p1.npz produced by executing the reference's
 *               k_impl / dkdT_impl (doc/demo/demo_nonlinear_heat_equation_part1.py:251-271)
 *               (tests/golden/make_golden_conductivity.py).
 *
 * The arithmetic follows the reference statement by statement (dense C_elas and `deviatoric`
 * mat-vecs, np.dot, np.outer), compiled with -ffp-contract=off so no FMA is introduced that NumPy
 * would not use. Remaining differences vs. NumPy are summation-order effects inside BLAS (<= a few ulp).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_MAX_D 6

/* demo_plasticity_von_mises.py:185-204: lmbda, mu, C_elas, deviatoric from (E, nu); d = 4 is the
 * reference's plane-strain Mandel vector, d = 6 the 3-D one built the same way. */
static void vm_constants(double E, double nu, int d, double* lmbda, double* mu, double C[ORACLE_MAX_D][ORACLE_MAX_D],
                         double dev[ORACLE_MAX_D][ORACLE_MAX_D]) {
    *lmbda = E * nu / (1.0 + nu) / (1.0 - 2.0 * nu);   /* :190 */
    *mu = E / 2.0 / (1.0 + nu);                        /* :191 */
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            C[i][j] = 0.0;
            dev[i][j] = (i == j) ? 1.0 : 0.0;          /* :203 np.eye */
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            C[i][j] = (i == j) ? *lmbda + 2.0 * *mu : *lmbda;   /* :193-199 */
            dev[i][j] -= 1.0 / 3.0;                              /* :204 */
        }
    for (int i = 3; i < d; ++i) C[i][i] = 2.0 * *mu;            /* :198 */
}

/* One quadrature point: `_kernel`, demo_plasticity_von_mises.py:307-326. */
static void vm_point(int d, const double C[ORACLE_MAX_D][ORACLE_MAX_D], const double dev[ORACLE_MAX_D][ORACLE_MAX_D],
                     double mu, double sigma_0, double H, const double* deps, const double* sigma_n, double p,
                     double* C_tang, double* sigma, double* dp_out) {
    double sigma_elastic[ORACLE_MAX_D], s[ORACLE_MAX_D], n_elas[ORACLE_MAX_D];
    for (int i = 0; i < d; ++i) {                       /* :309 sigma_n + C_elas @ deps */
        double acc = 0.0;
        for (int j = 0; j < d; ++j) acc += C[i][j] * deps[j];
        sigma_elastic[i] = sigma_n[i] + acc;
    }
    for (int i = 0; i < d; ++i) {                       /* :310 s = deviatoric @ sigma_elastic */
        double acc = 0.0;
        for (int j = 0; j < d; ++j) acc += dev[i][j] * sigma_elastic[j];
        s[i] = acc;
    }
    double ss = 0.0;
    for (int i = 0; i < d; ++i) ss += s[i] * s[i];
    const double sigma_eq = sqrt(3.0 / 2.0 * ss);                        /* :311 */
    const double f_elastic = sigma_eq - sigma_0 - H * p;                 /* :313 */
    const double f_elastic_plus = (f_elastic + sqrt(f_elastic * f_elastic)) / 2.0;  /* :314 */
    const double dp = f_elastic_plus / (3 * mu + H);                     /* :316 */
    for (int i = 0; i < d; ++i) n_elas[i] = s[i] / sigma_eq * f_elastic_plus / f_elastic;  /* :318 */
    const double beta = 3 * mu * dp / sigma_eq;                          /* :319 */
    for (int i = 0; i < d; ++i) sigma[i] = sigma_elastic[i] - beta * s[i];  /* :321 */
    const double a = 3 * mu * (3 * mu / (3 * mu + H) - beta);            /* :324 */
    const double b = 2 * mu * beta;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)                                      /* :323-324 */
            C_tang[i * d + j] = C[i][j] - a * (n_elas[i] * n_elas[j]) - b * dev[i][j];
    *dp_out = dp;
}

/* Batched caller: return_mapping, demo_plasticity_von_mises.py:298-332 (flat over cells x points).
 * prm = {E, nu, sigma_0, H}. nthreads <= 1: serial like the reference's Numba loop. */
int oracle_von_mises(const double* prm, int d, int64_t n, const double* deps, const double* sigma_n,
                     const double* p, double* C_tang, double* sigma, double* dp, int nthreads) {
    if (d != 4 && d != 6) return -2;
    double lmbda, mu, C[ORACLE_MAX_D][ORACLE_MAX_D], dev[ORACLE_MAX_D][ORACLE_MAX_D];
    vm_constants(prm[0], prm[1], d, &lmbda, &mu, C, dev);
    const double sigma_0 = prm[2], H = prm[3];
    (void)lmbda;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i)
        vm_point(d, C, dev, mu, sigma_0, H, deps + i * d, sigma_n + i * d, p[i], C_tang + i * d * d, sigma + i * d,
                 dp + i);
    return 0;
}

/* Nonlinear heat flux, demo_nonlinear_heat_equation_part2.py:215-261. Outputs may be NULL. */
int oracle_heat(double A, double B, int gdim, int64_t n, const double* T, const double* sigma, double* q,
                double* dqdT, double* dqdsigma, int nthreads) {
    if (gdim < 1 || gdim > 3) return -2;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        const double k = 1.0 / (A + B * T[i]);              /* :216 */
        for (int a = 0; a < gdim; ++a) {
            const double s = sigma[i * gdim + a];
            if (q) q[i * gdim + a] = -k * s;                /* :228 */
            if (dqdT) dqdT[i * gdim + a] = B * (k * k) * s; /* :246 */
            if (dqdsigma)
                for (int b = 0; b < gdim; ++b) dqdsigma[(i * gdim + a) * gdim + b] = -k * (a == b ? 1.0 : 0.0); /* :260 */
        }
    }
    return 0;
}

/* Scalar conductivity of the part-1 heat demo, demo_nonlinear_heat_equation_part1.py:251-271. Outputs may be NULL. */
int oracle_conductivity(double A, double B, int64_t n, const double* T, double* k_out, double* dkdT) {
    for (int64_t i = 0; i < n; ++i) {
        const double k = 1.0 / (A + B * T[i]);   /* :254 */
        if (k_out) k_out[i] = k;
        if (dkdT) dkdT[i] = -B * (k * k);        /* :271  -B * k_impl(T) ** 2 */
    }
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
