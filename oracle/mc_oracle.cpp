/*
 * mc_oracle.cpp — CPU restatement of the reference's Mohr-Coulomb return mapping and its
 * forward-mode-AD-through-the-Newton-loop tangent.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as dxo_oracle.c): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, link or call this.
 *
 * Reference: doc/demo/demo_plasticity_mohr_coulomb.py
 *   invariants, Abbo-Sloan K, surface, f, g   :282-391      residual r, drdy = jacfwd(r)   :420-462
 *   return_mapping (lax.while_loop Newton)     :469-533      dsigma_ddeps = jacfwd(return_mapping) :555
 * The reference obtains EVERY derivative by JAX forward-mode AD, including the consistent tangent,
 * which is jacfwd THROUGH the while_loop (the tangent of each Newton iterate is propagated, not the
 * implicit-function tangent at the converged point). This file restates that literally: the algorithm
 * is written once on a generic scalar type; derivatives come from nested dual numbers
 * (Dual<T> = forward-mode jvp along one seed direction), dgdsigma = 4 seeds over `surface`, drdy = 5
 * seeds over `r`, and the whole Newton loop — including the 5x5 linear solve — runs on a scalar that
 * carries the 4 tangents with respect to deps (DualN<4>), which is what jacfwd(return_mapping) does.
 *
 * Parity status: JAX is not installable here, so the oracle is pinned against
 * tests/golden/mohr_coulomb.npz, produced by EXECUTING THE REFERENCE'S OWN FUNCTION SOURCE
 * (AST-extracted, unmodified) under a thin jax/jnp shim backed by torch.func forward-mode AD
 * (tests/golden/make_golden_mohr_coulomb.py). Stated plainly: pinned to the reference source run on a
 * stand-in AD backend, not to JAX itself.
 */
#include <cmath>
#include <cstdint>
#include <cstring>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// forward-mode scalars
template <int N, class R = double>
struct DualN {  // value + N tangents (the deps directions); R = double is the oracle, R = long double the accuracy referee
    R v;
    R d[N];
    DualN() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
    DualN(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0.0; }  // NOLINT: implicit lift of constants
};

template <class T>
struct Dual {  // one jvp level on top of scalar type T
    T v, d;
    Dual() : v(0.0), d(0.0) {}
    Dual(double c) : v(c), d(0.0) {}  // NOLINT
    Dual(const T& v_, const T& d_) : v(v_), d(d_) {}
};

inline double primal(double x) { return x; }
template <int N, class R> inline double primal(const DualN<N, R>& x) { return (double)x.v; }
template <class T> inline double primal(const Dual<T>& x) { return primal(x.v); }

// ---- DualN algebra
template <int N, class R> inline DualN<N, R> operator+(const DualN<N, R>& a, const DualN<N, R>& b) { DualN<N, R> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N, class R> inline DualN<N, R> operator-(const DualN<N, R>& a, const DualN<N, R>& b) { DualN<N, R> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N, class R> inline DualN<N, R> operator-(const DualN<N, R>& a) { DualN<N, R> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N, class R> inline DualN<N, R> operator*(const DualN<N, R>& a, const DualN<N, R>& b) { DualN<N, R> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N, class R> inline DualN<N, R> operator/(const DualN<N, R>& a, const DualN<N, R>& b) { DualN<N, R> r; r.v = a.v / b.v; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int N, class R> inline DualN<N, R> sqrt(const DualN<N, R>& a) { DualN<N, R> r; r.v = std::sqrt(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / (R(2.0) * r.v); return r; }
template <int N, class R> inline DualN<N, R> asin(const DualN<N, R>& a) { DualN<N, R> r; r.v = std::asin(a.v); const R g = R(1.0) / std::sqrt(R(1.0) - a.v * a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g; return r; }
template <int N, class R> inline DualN<N, R> sin(const DualN<N, R>& a) { DualN<N, R> r; r.v = std::sin(a.v); const R g = std::cos(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g; return r; }
template <int N, class R> inline DualN<N, R> cos(const DualN<N, R>& a) { DualN<N, R> r; r.v = std::cos(a.v); const R g = -std::sin(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * g; return r; }

// ---- Dual<T> algebra (recursive)
template <class T> inline Dual<T> operator+(const Dual<T>& a, const Dual<T>& b) { return Dual<T>(a.v + b.v, a.d + b.d); }
template <class T> inline Dual<T> operator-(const Dual<T>& a, const Dual<T>& b) { return Dual<T>(a.v - b.v, a.d - b.d); }
template <class T> inline Dual<T> operator-(const Dual<T>& a) { return Dual<T>(-a.v, -a.d); }
template <class T> inline Dual<T> operator*(const Dual<T>& a, const Dual<T>& b) { return Dual<T>(a.v * b.v, a.d * b.v + a.v * b.d); }
template <class T> inline Dual<T> operator/(const Dual<T>& a, const Dual<T>& b) { T q = a.v / b.v; return Dual<T>(q, (a.d - q * b.d) / b.v); }
using std::asin;
using std::cos;
using std::sin;
using std::sqrt;
template <class T> inline Dual<T> sqrt(const Dual<T>& a) { T s = sqrt(a.v); return Dual<T>(s, a.d / (T(2.0) * s)); }
template <class T> inline Dual<T> asin(const Dual<T>& a) { return Dual<T>(asin(a.v), a.d / sqrt(T(1.0) - a.v * a.v)); }
template <class T> inline Dual<T> sin(const Dual<T>& a) { return Dual<T>(sin(a.v), a.d * cos(a.v)); }
template <class T> inline Dual<T> cos(const Dual<T>& a) { return Dual<T>(cos(a.v), -(a.d * sin(a.v))); }

// mixed with double constants (works for DualN and Dual<T> alike)
#define DXO_MIXED(OP)                                                                                   \
    template <int N, class R> inline DualN<N, R> operator OP(const DualN<N, R>& a, double b) { return a OP DualN<N, R>(b); } \
    template <int N, class R> inline DualN<N, R> operator OP(double a, const DualN<N, R>& b) { return DualN<N, R>(a) OP b; } \
    template <class T> inline Dual<T> operator OP(const Dual<T>& a, double b) { return a OP Dual<T>(b); }   \
    template <class T> inline Dual<T> operator OP(double a, const Dual<T>& b) { return Dual<T>(a) OP b; }
DXO_MIXED(+)
DXO_MIXED(-)
DXO_MIXED(*)
DXO_MIXED(/)
#undef DXO_MIXED

// jnp.clip(x, lo, hi): value clamped; tangent passes only strictly inside (the select-based jvp)
template <class T> inline T clip(const T& x, double lo, double hi) {
    const double p = primal(x);
    if (p < lo) return T(lo);
    if (p > hi) return T(hi);
    return x;
}

// ------------------------------------------------------------------------------------------------
struct McParams {  // mirrors dxo_mc_params (include/dxo.h)
    double E, nu, c, phi, psi, theta_T, a, tol;
    int32_t nitermax, pad;
};

struct McModel {
    McParams p;
    double lmbda, mu, C[4][4], dev[4][4], tr[4], coeff3;
    explicit McModel(const McParams& q) : p(q) {
        lmbda = p.E * p.nu / ((1.0 + p.nu) * (1.0 - 2.0 * p.nu));  // :405
        mu = p.E / (2.0 * (1.0 + p.nu));                            // :406
        const double Cm[4][4] = {{lmbda + 2 * mu, lmbda, lmbda, 0}, {lmbda, lmbda + 2 * mu, lmbda, 0},
                                 {lmbda, lmbda, lmbda + 2 * mu, 0}, {0, 0, 0, 2 * mu}};  // :407-415
        const double dv[4][4] = {{2.0 / 3.0, -1.0 / 3.0, -1.0 / 3.0, 0.0}, {-1.0 / 3.0, 2.0 / 3.0, -1.0 / 3.0, 0.0},
                                 {-1.0 / 3.0, -1.0 / 3.0, 2.0 / 3.0, 0.0}, {0.0, 0.0, 0.0, 1.0}};  // :352-360
        std::memcpy(C, Cm, sizeof C);
        std::memcpy(dev, dv, sizeof dev);
        tr[0] = tr[1] = tr[2] = 1.0;  // :361
        tr[3] = 0.0;
        coeff3 = 18.0 * std::cos(3.0 * p.theta_T) * std::cos(3.0 * p.theta_T) * std::cos(3.0 * p.theta_T);  // :310
    }

    // :282-283
    template <class T> T J3(const T* s) const { return s[2] * (s[0] * s[1] - s[3] * s[3] / 2.0); }
    // :286-287 (vdot)
    template <class T> T J2(const T* s) const { return 0.5 * (s[0] * s[0] + s[1] * s[1] + s[2] * s[2] + s[3] * s[3]); }
    // :290-295
    template <class T> T theta(const T* s) const {
        T J2_ = J2(s);
        T arg = -(3.0 * std::sqrt(3.0) * J3(s)) / (2.0 * sqrt(J2_ * J2_ * J2_));
        arg = clip(arg, -1.0, 1.0);
        return (1.0 / 3.0) * asin(arg);
    }
    static int sign(double x) { return x < 0.0 ? -1 : 1; }  // :298-299
    double coeff1(double angle) const { return std::cos(p.theta_T) - (1.0 / std::sqrt(3.0)) * std::sin(angle) * std::sin(p.theta_T); }  // :302-303
    double coeff2(double th, double angle) const { return sign(th) * std::sin(p.theta_T) + (1.0 / std::sqrt(3.0)) * std::sin(angle) * std::cos(p.theta_T); }  // :306-307
    double Cc(double th, double angle) const {  // :313-316
        return (-std::cos(3.0 * p.theta_T) * coeff1(angle) - 3.0 * sign(th) * std::sin(3.0 * p.theta_T) * coeff2(th, angle)) / coeff3;
    }
    double Bc(double th, double angle) const {  // :319-322
        return (sign(th) * std::sin(6.0 * p.theta_T) * coeff1(angle) - 6.0 * std::cos(6.0 * p.theta_T) * coeff2(th, angle)) / coeff3;
    }
    double Ac(double th, double angle) const {  // :325-331
        return -(1.0 / std::sqrt(3.0)) * std::sin(angle) * sign(th) * std::sin(p.theta_T) - Bc(th, angle) * sign(th) * std::sin(3 * p.theta_T) -
               Cc(th, angle) * std::sin(3.0 * p.theta_T) * std::sin(3.0 * p.theta_T) + std::cos(p.theta_T);
    }
    // :334-345
    template <class T> T K(const T& th, double angle) const {
        const double thp = primal(th);
        if (std::fabs(thp) > p.theta_T) {
            T s3 = sin(3.0 * th);
            return Ac(thp, angle) + Bc(thp, angle) * s3 + Cc(thp, angle) * s3 * s3;
        }
        return cos(th) - (1.0 / std::sqrt(3.0)) * std::sin(angle) * sin(th);
    }
    double a_g(double angle) const { return p.a * std::tan(p.phi) / std::tan(angle); }  // :348-349
    // :364-374
    template <class T> T surface(const T* sig, double angle) const {
        T s[4];
        for (int i = 0; i < 4; ++i) {
            T acc = dev[i][0] * sig[0];
            for (int j = 1; j < 4; ++j) acc = acc + dev[i][j] * sig[j];
            s[i] = acc;
        }
        T I1 = tr[0] * sig[0] + tr[1] * sig[1] + tr[2] * sig[2] + tr[3] * sig[3];
        T th = theta(s);
        T Kt = K(th, angle);
        return (I1 / 3.0 * std::sin(angle)) + sqrt(J2(s) * Kt * Kt + a_g(angle) * a_g(angle) * std::sin(angle) * std::sin(angle)) -
               p.c * std::cos(angle);
    }
    template <class T> T f(const T* sig) const { return surface(sig, p.phi); }  // :383-384
    template <class T> T g(const T* sig) const { return surface(sig, p.psi); }  // :387-388
    // dgdsigma = jax.jacfwd(g)  (:391): one jvp per basis direction
    template <class T> void dgdsigma(const T* sig, T* out) const {
        for (int i = 0; i < 4; ++i) {
            Dual<T> x[4];
            for (int j = 0; j < 4; ++j) x[j] = Dual<T>(sig[j], T(j == i ? 1.0 : 0.0));
            out[i] = g(x).d;
        }
    }
    template <class T> void C_times(const T* x, T* out) const {
        for (int i = 0; i < 4; ++i) {
            T acc = C[i][0] * x[0];
            for (int j = 1; j < 4; ++j) acc = acc + C[i][j] * x[j];
            out[i] = acc;
        }
    }
    // r(y, deps, sigma_n)  (:420-459). `yielding` is evaluated on the trial stress, as in :421-422/:439-440.
    template <class T> void r(const T* y, const T* deps, const T* sn, T* res) const {
        T Ce[4], trial[4];
        C_times(deps, Ce);
        for (int i = 0; i < 4; ++i) trial[i] = sn[i] + Ce[i];
        const bool elastic = primal(f(trial)) <= 0.0;
        const T& dlambda = y[4];
        T depsp[4];
        if (elastic) {
            for (int i = 0; i < 4; ++i) depsp[i] = T(0.0);           // :424-425
        } else {
            T gg[4];
            dgdsigma(y, gg);
            for (int i = 0; i < 4; ++i) depsp[i] = dlambda * gg[i];  // :427-428
        }
        T diff[4], Cd[4];
        for (int i = 0; i < 4; ++i) diff[i] = deps[i] - depsp[i];
        C_times(diff, Cd);
        for (int i = 0; i < 4; ++i) res[i] = y[i] - sn[i] - Cd[i];  // :435
        res[4] = elastic ? dlambda : f(y);                           // :442-448
    }
    // drdy = jax.jacfwd(r)  (:462)
    template <class T> void drdy(const T* y, const T* deps, const T* sn, T (*J)[5]) const {
        for (int j = 0; j < 5; ++j) {
            Dual<T> yy[5], dd[4], ss[4], rr[5];
            for (int i = 0; i < 5; ++i) yy[i] = Dual<T>(y[i], T(i == j ? 1.0 : 0.0));
            for (int i = 0; i < 4; ++i) {
                dd[i] = Dual<T>(deps[i], T(0.0));
                ss[i] = Dual<T>(sn[i], T(0.0));
            }
            r(yy, dd, ss, rr);
            for (int i = 0; i < 5; ++i) J[i][j] = rr[i].d;
        }
    }
};

template <class T> T norm5(const T* r) {  // jnp.linalg.norm
    T acc = r[0] * r[0];
    for (int i = 1; i < 5; ++i) acc = acc + r[i] * r[i];
    return sqrt(acc);
}

// jnp.linalg.solve(j, b): LU with partial pivoting (by primal magnitude), carried out on T
template <class T> void solve5(T (*A)[5], T* b, T* x) {
    int perm[5] = {0, 1, 2, 3, 4};
    for (int k = 0; k < 5; ++k) {
        int piv = k;
        double best = std::fabs(primal(A[perm[k]][k]));
        for (int i = k + 1; i < 5; ++i) {
            const double v = std::fabs(primal(A[perm[i]][k]));
            if (v > best) { best = v; piv = i; }
        }
        const int tmp = perm[k]; perm[k] = perm[piv]; perm[piv] = tmp;
        const int pk = perm[k];
        for (int i = k + 1; i < 5; ++i) {
            const int pi = perm[i];
            T m = A[pi][k] / A[pk][k];
            for (int j = k + 1; j < 5; ++j) A[pi][j] = A[pi][j] - m * A[pk][j];
            b[pi] = b[pi] - m * b[pk];
        }
    }
    for (int k = 4; k >= 0; --k) {
        const int pk = perm[k];
        T acc = b[pk];
        for (int j = k + 1; j < 5; ++j) acc = acc - A[pk][j] * x[j];
        x[k] = acc / A[pk][k];
    }
}

// return_mapping (:474-533) on scalar type T. Outputs: y (sigma, dlambda), niter, yielding, norm_res.
template <class T>
void return_mapping(const McModel& M, const T* deps, const T* sn, T* y, int* niter_out, double* yielding_out, double* norm_res_out) {
    int niter = 0;
    for (int i = 0; i < 4; ++i) y[i] = sn[i];  // :497-498
    y[4] = T(0.0);
    T res[5];
    M.r(y, deps, sn, res);
    T norm_res = norm5(res);
    const double norm_res0 = primal(norm_res);          // :501
    // cond_fun :503-505 — note 0/0 = NaN > tol is false: zero iterations when deps == 0
    while ((primal(norm_res) / norm_res0 > M.p.tol) && (niter < M.p.nitermax)) {
        T J[5][5], rhs[5], step[5];
        M.drdy(y, deps, sn, J);                          // :512
        for (int i = 0; i < 5; ++i) rhs[i] = -res[i];
        solve5(J, rhs, step);                            // :513
        for (int i = 0; i < 5; ++i) y[i] = y[i] + step[i];   // :514
        M.r(y, deps, sn, res);                           // :516
        norm_res = norm5(res);                           // :517
        niter += 1;                                      // :520
    }
    T Ce[4], trial[4];
    M.C_times(deps, Ce);
    for (int i = 0; i < 4; ++i) trial[i] = sn[i] + Ce[i];
    *yielding_out = primal(M.f(trial));                  // :530-531
    *niter_out = niter;
    *norm_res_out = primal(norm_res);
}

}  // namespace

extern "C" {

/* dsigma_ddeps_vec (:555, :574) over n points. prm points at a dxo_mc_params-compatible struct.
 * Outputs: C_tang [n][4][4] = d sigma / d deps through the loop, sigma [n][4]; optional diagnostics. */
int oracle_mohr_coulomb(const void* prm, int64_t n, const double* deps, const double* sigma_n, double* C_tang,
                        double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda, int nthreads) {
    McParams p;
    std::memcpy(&p, prm, sizeof p);
    const McModel M(p);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        typedef DualN<4> S;
        S e[4], s0[4], y[5];
        for (int k = 0; k < 4; ++k) {
            e[k] = S(deps[i * 4 + k]);
            e[k].d[k] = 1.0;  // jacfwd seeds: tangent basis of deps
            s0[k] = S(sigma_n[i * 4 + k]);
        }
        int it;
        double yl, nr;
        return_mapping<S>(M, e, s0, y, &it, &yl, &nr);
        for (int a = 0; a < 4; ++a) {
            sigma[i * 4 + a] = y[a].v;
            for (int b = 0; b < 4; ++b) C_tang[i * 16 + a * 4 + b] = y[a].d[b];
        }
        if (niter) niter[i] = it;
        if (yielding) yielding[i] = yl;
        if (norm_res) norm_res[i] = nr;
        if (dlambda) dlambda[i] = y[4].v;
    }
    return 0;
}

/* The same algorithm with 80-bit long double under the dual numbers: NOT the reference's arithmetic (JAX runs in fp64,
 * :103) but a referee for it — where the fp64 restatement and the HIP lane math disagree (they do, up to 1e-9 of the
 * tangent's scale, at the compression / extension meridians where the reference's sin(3 asin(arg)/3) chain cancels terms of
 * size (1 - arg^2)^(-5/2)), this tells which side carries the rounding error. Inputs and outputs are doubles. The Newton
 * loop's exit test uses the same tol on the long-double residual, so iteration counts can differ from the fp64 run for
 * residuals sitting at the tolerance; niter is returned so that the test can select points with equal counts. */
int oracle_mohr_coulomb_ld(const void* prm, int64_t n, const double* deps, const double* sigma_n, double* C_tang, double* sigma,
                           int32_t* niter, int nthreads) {
    McParams p;
    std::memcpy(&p, prm, sizeof p);
    const McModel M(p);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        typedef DualN<4, long double> S;
        S e[4], s0[4], y[5];
        for (int k = 0; k < 4; ++k) {
            e[k] = S(deps[i * 4 + k]);
            e[k].d[k] = 1.0L;
            s0[k] = S(sigma_n[i * 4 + k]);
        }
        int it;
        double yl, nr;
        return_mapping<S>(M, e, s0, y, &it, &yl, &nr);
        for (int a = 0; a < 4; ++a) {
            sigma[i * 4 + a] = (double)y[a].v;
            for (int b = 0; b < 4; ++b) C_tang[i * 16 + a * 4 + b] = (double)y[a].d[b];
        }
        if (niter) niter[i] = it;
    }
    return 0;
}

/* return_mapping only (no tangent), plain doubles — the yield-surface tracing use (:902-906). */
int oracle_mohr_coulomb_sigma(const void* prm, int64_t n, const double* deps, const double* sigma_n, double* sigma,
                              int32_t* niter, double* yielding, double* norm_res, double* dlambda, int nthreads) {
    McParams p;
    std::memcpy(&p, prm, sizeof p);
    const McModel M(p);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 1 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double y[5], yl, nr;
        int it;
        return_mapping<double>(M, deps + i * 4, sigma_n + i * 4, y, &it, &yl, &nr);
        for (int a = 0; a < 4; ++a) sigma[i * 4 + a] = y[a];
        if (niter) niter[i] = it;
        if (yielding) yielding[i] = yl;
        if (norm_res) norm_res[i] = nr;
        if (dlambda) dlambda[i] = y[4];
    }
    return 0;
}

/* f(sigma), g(sigma), dg/dsigma for tests (surface continuity at |theta| = theta_T etc.) */
int oracle_mc_surface(const void* prm, int64_t n, const double* sigma, double* f_out, double* g_out, double* dg_out) {
    McParams p;
    std::memcpy(&p, prm, sizeof p);
    const McModel M(p);
    for (int64_t i = 0; i < n; ++i) {
        if (f_out) f_out[i] = M.f(sigma + i * 4);
        if (g_out) g_out[i] = M.g(sigma + i * 4);
        if (dg_out) M.dgdsigma(sigma + i * 4, dg_out + i * 4);
    }
    return 0;
}

}  // extern "C"
