// mc_core.h — per-point Mohr-Coulomb (Abbo-Sloan) return mapping with the forward-mode-through-Newton
// tangent, written for one GPU lane. Host/device compilable (the host build exists only so the math can
// be unit-tested against the oracle without a GPU; the product path is mohr_coulomb.hip).
//
// Reference: doc/demo/demo_plasticity_mohr_coulomb.py — surface/f/g :282-391, r/drdy :420-462,
// return_mapping :474-533, dsigma_ddeps = jacfwd(return_mapping) :555.
//
// The reference gets every derivative from JAX forward-mode AD: dg/dsigma inside r, dr/dy inside the
// Newton loop, and the tangent by differentiating THROUGH the loop. A GPU lane cannot afford nested
// dual numbers (a triple-nested 4-direction dual is 20+ doubles per scalar), so the same quantities are
// obtained in closed form:
//
//  1. g(sigma) = sin(a) I1/3 + F(J2, J3) - c cos(a),  F = sqrt(J2 K(theta)^2 + a_g^2 sin^2 a),
//     theta = asin(clip(-3 sqrt3 J3 / (2 J2^1.5)))/3. F is evaluated ONCE per iterate on a truncated Taylor
//     type in the two invariants (T23: all partials of F up to third order, 10 numbers). Tensor
//     derivatives follow from the chain rule with grad J2 = s, hess J2 = dev, grad/hess J3 polynomial in s.
//  2. Newton step on y = (sigma, dlambda) in compliance form: with S = C^-1, M = S + dlambda H_g (SPD),
//     J^-1 reduces to solves with M (LDL^T, no pivoting, fully unrolled) plus a scalar Schur complement.
//  3. Tangent recursion of jacfwd-through-while_loop (Y_k = d y_k / d deps, Y_0 = 0, step t = J^-1 r):
//         Y_{k+1} = J^-1 ( [C; 0] + (D J [Y_k]) t ),
//     (D J[v, dl] t)_sigma  = C ( dl H t_s + dlambda T(t_s) v + t_l H v ),   T(t) = D_t H (third derivatives)
//     (D J[v, dl] t)_lambda = (H_f t_s) . v
//     which is exactly what forward-mode AD of `y + solve(drdy(y), -r(y))` computes.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DXO_HD __host__ __device__ __forceinline__
#else
#define DXO_HD inline
#endif

namespace mc {

struct Const {  // derived once on the host from dxo_mc_params
    double lmbda, mu2;              // C_elas = lmbda 1(x)1 + 2 mu I (Mandel), :405-415
    double inv_E, nu;               // S = C^-1: (1/E) [[1,-nu,-nu,0],...,[0,0,0,1+nu]]
    double c, theta_T, tol;
    double sin3T;                   // sin(3 theta_T): |theta| > theta_T  <=>  |arg| > sin3T (asin is monotone)
    int32_t nitermax, same_angle;   // same_angle: phi == psi -> f and g share every derivative
    // per angle (index 0: phi -> f, 1: psi -> g)
    double sin_a[2], cos_a[2], k_lin[2];       // k_lin = sin(a)/sqrt(3)
    double ag2s2[2];                           // a_g(a)^2 sin^2 a, :348-349, :371
    double A[2][2], B[2][2], Cc[2][2];         // Abbo-Sloan coefficients for sign(theta) = -1 / +1, :302-331
};

// ---- truncated Taylor type in (x, y) = (J2, J3): f, fx, fy, fxx, fxy, fyy, fxxx, fxxy, fxyy, fyyy
struct T23 { double c[10]; };

DXO_HD T23 t_mul(const T23& u, const T23& v) {
    T23 w;
    w.c[0] = u.c[0] * v.c[0];
    w.c[1] = u.c[1] * v.c[0] + u.c[0] * v.c[1];
    w.c[2] = u.c[2] * v.c[0] + u.c[0] * v.c[2];
    w.c[3] = u.c[3] * v.c[0] + 2.0 * u.c[1] * v.c[1] + u.c[0] * v.c[3];
    w.c[4] = u.c[4] * v.c[0] + u.c[1] * v.c[2] + u.c[2] * v.c[1] + u.c[0] * v.c[4];
    w.c[5] = u.c[5] * v.c[0] + 2.0 * u.c[2] * v.c[2] + u.c[0] * v.c[5];
    w.c[6] = u.c[6] * v.c[0] + 3.0 * (u.c[3] * v.c[1] + u.c[1] * v.c[3]) + u.c[0] * v.c[6];
    w.c[7] = u.c[7] * v.c[0] + u.c[3] * v.c[2] + 2.0 * (u.c[4] * v.c[1] + u.c[1] * v.c[4]) + u.c[2] * v.c[3] + u.c[0] * v.c[7];
    w.c[8] = u.c[8] * v.c[0] + u.c[5] * v.c[1] + 2.0 * (u.c[4] * v.c[2] + u.c[2] * v.c[4]) + u.c[1] * v.c[5] + u.c[0] * v.c[8];
    w.c[9] = u.c[9] * v.c[0] + 3.0 * (u.c[5] * v.c[2] + u.c[2] * v.c[5]) + u.c[0] * v.c[9];
    return w;
}

// w = h(u) given h and its first three derivatives at u.c[0] (Faa di Bruno to third order)
DXO_HD T23 t_compose(const T23& u, double h0, double h1, double h2, double h3) {
    T23 w;
    const double ux = u.c[1], uy = u.c[2];
    w.c[0] = h0;
    w.c[1] = h1 * ux;
    w.c[2] = h1 * uy;
    w.c[3] = h1 * u.c[3] + h2 * ux * ux;
    w.c[4] = h1 * u.c[4] + h2 * ux * uy;
    w.c[5] = h1 * u.c[5] + h2 * uy * uy;
    w.c[6] = h1 * u.c[6] + 3.0 * h2 * u.c[3] * ux + h3 * ux * ux * ux;
    w.c[7] = h1 * u.c[7] + h2 * (u.c[3] * uy + 2.0 * u.c[4] * ux) + h3 * ux * ux * uy;
    w.c[8] = h1 * u.c[8] + h2 * (u.c[5] * ux + 2.0 * u.c[4] * uy) + h3 * ux * uy * uy;
    w.c[9] = h1 * u.c[9] + 3.0 * h2 * u.c[5] * uy + h3 * uy * uy * uy;
    return w;
}

DXO_HD T23 t_scale(const T23& u, double k) {
    T23 w;
    for (int i = 0; i < 10; ++i) w.c[i] = k * u.c[i];
    return w;
}

// 1 / sqrt(x) to fp64 rounding: the hardware's v_rsq_f64 and two Newton steps (9 instructions); sqrt(x) = x y, 1 / x = y y and
// x^-1.5 = y y y then cost one multiplication each — the library's sqrt and division are 14 and 12 instructions with long
// dependent chains, and the pass needs them in threes. (x = 0, inf, nan, < 0 give the nan / inf combinations the callers treat
// as "not a number" anyway: a hydrostatic stress has f = nan in the reference too.)
DXO_HD double mc_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(x);
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-(x * y), y, 1.0);
        y = fma(y * 0.5, e, y);
    }
    return y;
#else
    return 1.0 / sqrt(x);
#endif
}

// 1 / x to fp64 rounding (v_rcp_f64 and two Newton steps, 5 instructions; the library division is 12 with the scaling it needs
// for operands near the ends of the exponent range, which pivots of an SPD matrix of moduli are not)
DXO_HD double mc_recip(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    for (int it = 0; it < 2; ++it) y = fma(fma(-x, y, 1.0), y, y);
    return y;
#else
    return 1.0 / x;
#endif
}

DXO_HD T23 t_sqrt(const T23& u) {
    const double y = mc_rsqrt(u.c[0]);
    const double h0 = u.c[0] * y;         // sqrt u
    const double iu = y * y;              // 1 / u
    const double h1 = 0.5 * y;            // 1 / (2 sqrt u)
    const double h2 = -0.5 * h1 * iu;
    const double h3 = -1.5 * h2 * iu;
    return t_compose(u, h0, h1, h2, h3);
}

// ---- 4-vector helpers on the Mandel vector (xx, yy, zz, sqrt2 xy)
DXO_HD void devv(const double* v, double* out) {  // dev @ v, :352-360
    // no FMA contraction here: a hydrostatic stress must give s == 0 exactly (then J2 == 0 and f is NaN, as in
    // the reference); fma(-(v0+v1+v2), 1/3, v0) would leave a 1e-17 residue instead
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const double m = (v[0] + v[1] + v[2]) * (1.0 / 3.0);
    out[0] = v[0] - m;
    out[1] = v[1] - m;
    out[2] = v[2] - m;
    out[3] = v[3];
}
DXO_HD double dot4(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
DXO_HD void C_times(const Const& k, const double* x, double* out) {  // C_elas @ x
    const double t = k.lmbda * (x[0] + x[1] + x[2]);
    out[0] = t + k.mu2 * x[0];
    out[1] = t + k.mu2 * x[1];
    out[2] = t + k.mu2 * x[2];
    out[3] = k.mu2 * x[3];
}
DXO_HD void S_times(const Const& k, const double* x, double* out) {  // C_elas^-1 @ x
    const double t = k.nu * (x[0] + x[1] + x[2]);
    out[0] = k.inv_E * ((1.0 + k.nu) * x[0] - t);
    out[1] = k.inv_E * ((1.0 + k.nu) * x[1] - t);
    out[2] = k.inv_E * ((1.0 + k.nu) * x[2] - t);
    out[3] = k.inv_E * (1.0 + k.nu) * x[3];
}
// hess_s J3 (s) applied to w; linear in s (so it also serves D_t hess J3 with s -> dev t)
DXO_HD void Qs_times(const double* s, const double* w, double* out) {
    out[0] = s[2] * w[1] + s[1] * w[2];
    out[1] = s[2] * w[0] + s[0] * w[2];
    out[2] = s[1] * w[0] + s[0] * w[1] - s[3] * w[3];
    out[3] = -s[3] * w[2] - s[2] * w[3];
}

// Everything the Newton step needs about the yield surface / plastic potential at one stress state.
struct Surf {
    double s[4], q[4];     // grad J2 = s, grad J3 = dev q_s
    double I1;
    double f, g;           // f(sigma), g(sigma)
    T23 F[2];              // F for angle phi (f) and psi (g)
};

// The Lode argument arg(J2, J3) = -(3 sqrt3 / 2) J3 J2^(-3/2) as a Taylor object (:290-295, jnp.clip included), and
// theta = asin(arg) / 3 as a plain number. K(theta) is then composed with arg DIRECTLY in both branches:
//   rounded  (|theta| > theta_T):  K = A + B arg + C arg^2          (sin 3 theta = sin(asin(arg)) = arg)
//   plain    (|theta| <= theta_T): K = cos theta - k_lin sin theta,  theta = asin(arg) / 3
// so every point needs ONE third-order composition (Faa di Bruno) for K instead of one for theta plus two for
// cos / sin, and the two branches differ only in four scalars (K and its first three derivatives with respect to arg).
// In the rounded branch the derivatives of asin — which grow like (1 - arg^2)^(-1/2, -3/2, -5/2) towards the compression /
// extension meridians and cancel against cos(3 theta) -> 0 — never appear (composing asin and sin numerically there cost
// up to 1e-2 of the tangent at 1 - |arg| ~ 1e-5); in the plain branch |arg| <= sin(3 theta_T) < 1 keeps them tame.
struct LodeArg {
    T23 a;                  // arg and its partials in (J2, J3)
    double sn, cs;          // sin theta, cos theta, theta = asin(arg) / 3 (plain branch only)
    double t1, t2, t3;      // d theta / d arg, second, third derivative (plain branch only)
    bool rounded;
};

// sin(theta) and cos(theta) for theta = asin(u) / 3, |u| <= 1, given v = sqrt((1 - u)(1 + u)) = cos(3 theta) — WITHOUT asin, sin
// and cos (about 280 fp64 instructions on the GPU, a 55-deep dependent chain in a pass that is bound by exactly that):
// sin(theta) is the root of 3x - 4x^3 = u in [-1/2, 1/2] and cos(theta) the root of 4c^3 - 3c = v in [sqrt(3)/2, 1]. Newton on
// whichever cubic is well conditioned (|u| <= 1/2: the sine's, derivative >= 2.6; else the cosine's, derivative >= 6): a cubic
// fp32 seed (2e-5), one fp32 step (5e-8), two fp64 steps with the derivative's reciprocal taken in fp32 (its 1e-7 error only
// slows the convergence: 5e-8 -> 6e-15 -> fp64 rounding). The other function follows from the triple-angle identities without
// cancellation: cos = v / (1 - 4 sin^2), sin = u / (4 cos^2 - 1). Against a 40-digit evaluation both are within 3e-16 relative
// over the whole range (the library chain: 3e-16), including u -> 1 where asin itself is ill-conditioned.
DXO_HD float mc_rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
DXO_HD void lode_sin_cos(double u, double v, double& sn, double& cs) {
    const double au = fabs(u);
    const bool cb = au > 0.5;
    const double rhs = cb ? v : au;
    const double a1 = cb ? -3.0 : 3.0, a3 = cb ? 4.0 : -4.0;
    const float rf = (float)rhs, a1f = cb ? -3.0f : 3.0f, a3f = cb ? 4.0f : -4.0f;
    const float r2 = rf * rf;
    float z = cb ? (0.8660486f + rf * (0.16574173f + rf * (-0.0420846f + rf * 0.01047643f)))
                 : rf * (0.33336093f + r2 * (0.04846038f + r2 * 0.02903655f));
    {
        const float t = z * z;
        z -= (z * (a1f + a3f * t) - rf) * mc_rcp_f32(a1f + 3.0f * a3f * t);
    }
    double zd = (double)z;
    for (int it = 0; it < 2; ++it) {
        const double t = zd * zd;
        const double f = fma(zd, fma(a3, t, a1), -rhs);
        const double rd = (double)mc_rcp_f32((float)fma(3.0 * a3, t, a1));
        zd = fma(-f, rd, zd);
    }
    const double t = zd * zd;
    const double other = (cb ? au : v) * mc_recip(cb ? fma(4.0, t, -1.0) : fma(-4.0, t, 1.0));
    sn = copysign(cb ? other : zd, u);
    cs = cb ? zd : other;
}

DXO_HD void lode_arg(const Const& k, double J2, double J3, LodeArg& o) {
    const double y2 = mc_rsqrt(J2);
    const double iJ2 = y2 * y2;
    const double h0 = iJ2 * y2;         // J2^-1.5 (the same expression as in f_value)
    const double h1 = -1.5 * h0 * iJ2;
    const double h2 = -2.5 * h1 * iJ2;
    const double h3 = -3.5 * h2 * iJ2;
    const double kk = -(3.0 * sqrt(3.0)) / 2.0;
    T23& a = o.a;
    a.c[0] = kk * J3 * h0;
    a.c[1] = kk * J3 * h1;
    a.c[2] = kk * h0;
    a.c[3] = kk * J3 * h2;
    a.c[4] = kk * h1;
    a.c[5] = 0.0;
    a.c[6] = kk * J3 * h3;
    a.c[7] = kk * h2;
    a.c[8] = 0.0;
    a.c[9] = 0.0;
    if (a.c[0] < -1.0 || a.c[0] > 1.0) {  // jnp.clip: constant outside, no tangent
        const double v = a.c[0] < 0.0 ? -1.0 : 1.0;
        for (int i = 1; i < 10; ++i) a.c[i] = 0.0;
        a.c[0] = v;
    }
    const double u = a.c[0];
    o.rounded = fabs(u) > k.sin3T;
    // (locals, not references into `o`: with the struct's fields written under the branch the compiler kept part of it in scratch)
    double sn = 0.0, cs = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (!o.rounded) {
        const double x1 = (1.0 - u) * (1.0 + u);
        const double w = mc_rsqrt(x1);      // 1 / cos(3 theta)
        lode_sin_cos(u, x1 * w, sn, cs);
        const double w3 = w * w * w;
        t1 = w * (1.0 / 3.0);
        t2 = u * w3 * (1.0 / 3.0);
        t3 = (w3 + 3.0 * u * u * w3 * w * w) * (1.0 / 3.0);
    }
    o.sn = sn; o.cs = cs; o.t1 = t1; o.t2 = t2; o.t3 = t3;
}

// F(J2, J3) = sqrt(J2 K(theta)^2 + a_g^2 sin^2 a) for angle index ia, :334-345, :364-374
DXO_HD T23 F_taylor(const Const& k, int ia, double J2, const LodeArg& L) {
    double k0, k1, k2, k3;   // K and dK/darg, d2K/darg2, d3K/darg3
    const double u = L.a.c[0];
    if (L.rounded) {
        const bool neg = u < 0.0;        // sign(theta) = sign(arg), :298-299; selects, not indexing: the constants live in scalar
        const double Ak = neg ? k.A[ia][0] : k.A[ia][1], Bk = neg ? k.B[ia][0] : k.B[ia][1], Ck = neg ? k.Cc[ia][0] : k.Cc[ia][1];   // registers
        k0 = Ak + (Bk + Ck * u) * u;
        k1 = Bk + 2.0 * Ck * u;
        k2 = 2.0 * Ck;
        k3 = 0.0;
    } else {
        const double kl = k.k_lin[ia];
        const double d1 = -L.sn - kl * L.cs;   // dK/dtheta
        const double d2 = -L.cs + kl * L.sn;   // d2K/dtheta2;  d3K/dtheta3 = -d1
        k0 = L.cs - kl * L.sn;
        k1 = d1 * L.t1;
        k2 = d2 * L.t1 * L.t1 + d1 * L.t2;
        k3 = -d1 * L.t1 * L.t1 * L.t1 + 3.0 * d2 * L.t1 * L.t2 + d1 * L.t3;
    }
    const T23 K = t_compose(L.a, k0, k1, k2, k3);
    const T23 KK = t_compose(K, K.c[0] * K.c[0], 2.0 * K.c[0], 2.0, 0.0);   // K^2
    T23 w;  // x * KK + const, x = J2 (dx = 1)
    w.c[0] = J2 * KK.c[0] + k.ag2s2[ia];
    w.c[1] = KK.c[0] + J2 * KK.c[1];
    w.c[2] = J2 * KK.c[2];
    w.c[3] = 2.0 * KK.c[1] + J2 * KK.c[3];
    w.c[4] = KK.c[2] + J2 * KK.c[4];
    w.c[5] = J2 * KK.c[5];
    w.c[6] = 3.0 * KK.c[3] + J2 * KK.c[6];
    w.c[7] = 2.0 * KK.c[4] + J2 * KK.c[7];
    w.c[8] = KK.c[5] + J2 * KK.c[8];
    w.c[9] = J2 * KK.c[9];
    return t_sqrt(w);
}

// SAME (phi == psi, the reference's demo): f and g share every derivative, F[0] and f are never formed or read — as a
// compile-time fact, so the ten doubles of F[0] cost no registers in the Newton kernel.
template <bool SAME>
DXO_HD void surf_eval(const Const& k, const double* sig, Surf& o) {
    devv(sig, o.s);
    o.I1 = sig[0] + sig[1] + sig[2];
    const double* s = o.s;
    const double J2 = 0.5 * dot4(s, s);                       // :286-287
    const double J3 = s[2] * (s[0] * s[1] - s[3] * s[3] / 2.0);  // :282-283
    const double qs[4] = {s[1] * s[2], s[0] * s[2], s[0] * s[1] - s[3] * s[3] / 2.0, -s[2] * s[3]};
    devv(qs, o.q);
    LodeArg lode;
    lode_arg(k, J2, J3, lode);
    o.F[1] = F_taylor(k, 1, J2, lode);
    o.g = o.I1 / 3.0 * k.sin_a[1] + o.F[1].c[0] - k.c * k.cos_a[1];
    if constexpr (SAME) {
        o.f = o.g;
    } else {
        o.F[0] = F_taylor(k, 0, J2, lode);
        o.f = o.I1 / 3.0 * k.sin_a[0] + o.F[0].c[0] - k.c * k.cos_a[0];
    }
}

DXO_HD void grad_surface(const Const& k, const Surf& e, int ia, double* out) {  // d surface / d sigma
    const double Fx = e.F[ia].c[1], Fy = e.F[ia].c[2];
    const double h = k.sin_a[ia] / 3.0;
    out[0] = h + Fx * e.s[0] + Fy * e.q[0];
    out[1] = h + Fx * e.s[1] + Fy * e.q[1];
    out[2] = h + Fx * e.s[2] + Fy * e.q[2];
    out[3] = Fx * e.s[3] + Fy * e.q[3];
}

// H v = hess(surface) v
DXO_HD void hess_apply(const Surf& e, int ia, const double* v, double* out) {
    const double* F = e.F[ia].c;
    double Pv[4], Qv[4], tmp[4];
    devv(v, Pv);
    Qs_times(e.s, Pv, tmp);
    devv(tmp, Qv);
    const double al = dot4(e.s, v), be = dot4(e.q, v);
    const double cp = F[3] * al + F[4] * be, cq = F[4] * al + F[5] * be;
    for (int i = 0; i < 4; ++i) out[i] = F[1] * Pv[i] + F[2] * Qv[i] + cp * e.s[i] + cq * e.q[i];
}

// Precomputed pieces of T(t) = D_t hess(surface) for a fixed direction t
struct Third {
    double Pt[4], Qt[4], dPt[4];  // dev t, (hess J3) t, dev t again for R_t
    double a, b;                  // s.t, q.t
    double dFx, dFy, dFxx, dFxy, dFyy;
};
DXO_HD void third_setup(const Surf& e, int ia, const double* t, Third& o) {
    const double* F = e.F[ia].c;
    double tmp[4];
    devv(t, o.Pt);
    Qs_times(e.s, o.Pt, tmp);
    devv(tmp, o.Qt);
    o.a = dot4(e.s, t);
    o.b = dot4(e.q, t);
    o.dFx = F[3] * o.a + F[4] * o.b;
    o.dFy = F[4] * o.a + F[5] * o.b;
    o.dFxx = F[6] * o.a + F[7] * o.b;
    o.dFxy = F[7] * o.a + F[8] * o.b;
    o.dFyy = F[8] * o.a + F[9] * o.b;
}
// T(t) v = D_t (H v)
DXO_HD void third_apply(const Surf& e, int ia, const Third& T, const double* v, double* out) {
    const double* F = e.F[ia].c;
    double Pv[4], Qv[4], Rv[4], tmp[4];
    devv(v, Pv);
    Qs_times(e.s, Pv, tmp);
    devv(tmp, Qv);
    Qs_times(T.Pt, Pv, tmp);  // hess_s J3 is linear in s: D_t -> s replaced by dev t
    devv(tmp, Rv);
    const double al = dot4(e.s, v), be = dot4(e.q, v);
    const double dal = dot4(T.Pt, v), dbe = dot4(T.Qt, v);
    const double cp = T.dFxx * al + F[3] * dal + T.dFxy * be + F[4] * dbe;
    const double cq = T.dFxy * al + F[4] * dal + T.dFyy * be + F[5] * dbe;
    const double ep = F[3] * al + F[4] * be, eq = F[4] * al + F[5] * be;
    for (int i = 0; i < 4; ++i)
        out[i] = T.dFx * Pv[i] + T.dFy * Qv[i] + F[2] * Rv[i] + cp * e.s[i] + ep * T.Pt[i] + cq * e.q[i] + eq * T.Qt[i];
}

// ---- the same two operators as dense symmetric matrices (10 numbers: 00 01 02 03 11 12 13 22 23 33)
// A pass applies hess(g) to six vectors (four unit vectors for M, the step, four tangent columns... ) and T(t) to four: formed
// ONCE as matrices — hess = Fx P + Fy B(s) + [s q] [[Fxx Fxy][Fxy Fyy]] [s q]^T with P = dev and B(s) = dev Q(s) dev — they cost
// 60 + 100 flops and every application 16 FMAs, instead of 60 / 110 flops per application of the structured forms above.
// For deviatoric s (s0 + s1 + s2 = 0):  3 B(s) = [[2 s0, 2 s2, 2 s1, s3], [., 2 s1, 2 s0, s3], [., ., 2 s2, -2 s3], [., ., ., -3 s2]].
struct Sym4 { double m[10]; };
DXO_HD void sym_apply(const Sym4& A, const double* v, double* out) {
    out[0] = A.m[0] * v[0] + A.m[1] * v[1] + A.m[2] * v[2] + A.m[3] * v[3];
    out[1] = A.m[1] * v[0] + A.m[4] * v[1] + A.m[5] * v[2] + A.m[6] * v[3];
    out[2] = A.m[2] * v[0] + A.m[5] * v[1] + A.m[7] * v[2] + A.m[8] * v[3];
    out[3] = A.m[3] * v[0] + A.m[6] * v[1] + A.m[8] * v[2] + A.m[9] * v[3];
}
DXO_HD void dev_q_dev3(const double* s, double* B) {   // 3 B(s)
    B[0] = 2.0 * s[0]; B[1] = 2.0 * s[2]; B[2] = 2.0 * s[1]; B[3] = s[3];
    B[4] = 2.0 * s[1]; B[5] = 2.0 * s[0]; B[6] = s[3];
    B[7] = 2.0 * s[2]; B[8] = -2.0 * s[3];
    B[9] = -3.0 * s[2];
}
// x_i y_j + z_i w_j for i <= j, added to A (the caller's sum is symmetric as a whole)
DXO_HD void sym_add_outer2(Sym4& A, const double* x, const double* y, const double* z, const double* w) {
    int n = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = i; j < 4; ++j, ++n) A.m[n] += x[i] * y[j] + z[i] * w[j];
}
DXO_HD void sym_add_dev(Sym4& A, double c) {   // A += c P,  P = dev
    const double d = c * (2.0 / 3.0), o = c * (-1.0 / 3.0);
    A.m[0] += d; A.m[1] += o; A.m[2] += o;
    A.m[4] += d; A.m[5] += o;
    A.m[7] += d;
    A.m[9] += c;
}
// hess(surface) dense; also returns a = Fxx s + Fxy q, b = Fxy s + Fyy q (third_dense needs them again)
DXO_HD void hess_dense(const Surf& e, int ia, Sym4& H, double* a, double* b) {
    const double* F = e.F[ia].c;
    for (int i = 0; i < 4; ++i) {
        a[i] = F[3] * e.s[i] + F[4] * e.q[i];
        b[i] = F[4] * e.s[i] + F[5] * e.q[i];
    }
    double B[10];
    dev_q_dev3(e.s, B);
    const double fy3 = F[2] * (1.0 / 3.0);
    for (int n = 0; n < 10; ++n) H.m[n] = fy3 * B[n];
    sym_add_dev(H, F[1]);
    sym_add_outer2(H, a, e.s, b, e.q);
}
// T(t) = D_t hess(surface) dense, for the direction t
DXO_HD void third_dense(const Surf& e, int ia, const double* t, const double* a, const double* b, Sym4& T) {
    const double* F = e.F[ia].c;
    double Pt[4], Qt[4], B[10], Bt[10];
    devv(t, Pt);
    dev_q_dev3(e.s, B);
    {
        Sym4 Bs;
        for (int n = 0; n < 10; ++n) Bs.m[n] = B[n] * (1.0 / 3.0);
        sym_apply(Bs, t, Qt);   // dev Q(s) dev t
    }
    dev_q_dev3(Pt, Bt);
    const double al = dot4(e.s, t), be = dot4(e.q, t);
    const double dFx = F[3] * al + F[4] * be, dFy = F[4] * al + F[5] * be;
    const double dFxx = F[6] * al + F[7] * be, dFxy = F[7] * al + F[8] * be, dFyy = F[8] * al + F[9] * be;
    double c1[4], c2[4];
    for (int i = 0; i < 4; ++i) {
        c1[i] = dFxx * e.s[i] + dFxy * e.q[i] + F[3] * Pt[i] + F[4] * Qt[i];
        c2[i] = dFxy * e.s[i] + dFyy * e.q[i] + F[4] * Pt[i] + F[5] * Qt[i];
    }
    const double dfy3 = dFy * (1.0 / 3.0), fy3 = F[2] * (1.0 / 3.0);
    for (int n = 0; n < 10; ++n) T.m[n] = dfy3 * B[n] + fy3 * Bt[n];
    sym_add_dev(T, dFx);
    sym_add_outer2(T, e.s, c1, e.q, c2);
    sym_add_outer2(T, Pt, a, Qt, b);
}

// LDL^T of a symmetric 4x4 (lower triangle in m[10]: 00,10,11,20,21,22,30,31,32,33), no pivoting
// (the four pivots are inverted once: a pass does 6 solves, and an fp64 division is ~12 instructions)
struct Ldl { double l10, l20, l21, l30, l31, l32, i0, i1, i2, i3; };
DXO_HD void ldl_factor(const Sym4& M, Ldl& f) {   // M.m: 00 01 02 03 11 12 13 22 23 33
    const double d0 = M.m[0];
    f.i0 = mc_recip(d0);
    f.l10 = M.m[1] * f.i0;
    f.l20 = M.m[2] * f.i0;
    f.l30 = M.m[3] * f.i0;
    const double d1 = M.m[4] - f.l10 * f.l10 * d0;
    f.i1 = mc_recip(d1);
    f.l21 = (M.m[5] - f.l20 * f.l10 * d0) * f.i1;
    f.l31 = (M.m[6] - f.l30 * f.l10 * d0) * f.i1;
    const double d2 = M.m[7] - f.l20 * f.l20 * d0 - f.l21 * f.l21 * d1;
    f.i2 = mc_recip(d2);
    f.l32 = (M.m[8] - f.l30 * f.l20 * d0 - f.l31 * f.l21 * d1) * f.i2;
    const double d3 = M.m[9] - f.l30 * f.l30 * d0 - f.l31 * f.l31 * d1 - f.l32 * f.l32 * d2;
    f.i3 = mc_recip(d3);
}
DXO_HD void ldl_solve(const Ldl& f, const double* b, double* x) {
    const double z0 = b[0];
    const double z1 = b[1] - f.l10 * z0;
    const double z2 = b[2] - f.l20 * z0 - f.l21 * z1;
    const double z3 = b[3] - f.l30 * z0 - f.l31 * z1 - f.l32 * z2;
    const double w3 = z3 * f.i3;
    const double w2 = z2 * f.i2 - f.l32 * w3;
    const double w1 = z1 * f.i1 - f.l21 * w2 - f.l31 * w3;
    const double w0 = z0 * f.i0 - f.l10 * w1 - f.l20 * w2 - f.l30 * w3;
    x[0] = w0; x[1] = w1; x[2] = w2; x[3] = w3;
}

struct Result {
    double sigma[4];
    double C_tang[16];  // row-major d sigma_i / d deps_j
    int32_t niter;
    double yielding, norm_res, dlambda;
};

// residual r(y) (:451-459) for the plastic branch, also returns rho = S r_sigma (compliance form)
DXO_HD double residual(const Const& k, const Surf& e, const double* sig, double dl, const double* deps, const double* sn,
                       const double* gradg, double* r_sig, double* r_f) {
    double de[4], Cd[4];
    for (int i = 0; i < 4; ++i) de[i] = deps[i] - dl * gradg[i];   // deps - deps_p, :427-435
    C_times(k, de, Cd);
    double acc = 0.0;
    for (int i = 0; i < 4; ++i) {
        r_sig[i] = sig[i] - sn[i] - Cd[i];
        acc += r_sig[i] * r_sig[i];
    }
    *r_f = e.f;                                                     // :446
    acc += e.f * e.f;
    return sqrt(acc);
}

// f(sigma) only — the trial-stress test (:422, :440, :531). Same expressions as the value path of
// surf_eval / F_taylor, without carrying derivatives.
// `ia`: which angle's constants to read. f belongs to phi (index 0); a kernel compiled for phi == psi passes 1 — the same
// numbers — so that only ONE set of per-angle constants is live in its scalar registers.
DXO_HD double f_value(const Const& k, const double* sig, int ia = 0) {
    double s[4];
    devv(sig, s);
    const double I1 = sig[0] + sig[1] + sig[2];
    const double J2 = 0.5 * dot4(s, s);
    const double J3 = s[2] * (s[0] * s[1] - s[3] * s[3] / 2.0);
    const double y2 = mc_rsqrt(J2);
    double arg = (-(3.0 * sqrt(3.0)) / 2.0) * J3 * ((y2 * y2) * y2);
    if (arg < -1.0 || arg > 1.0) arg = arg < 0.0 ? -1.0 : 1.0;
    double K;
    if (fabs(arg) > k.sin3T) {
        const bool neg = arg < 0.0;
        const double Ak = neg ? k.A[ia][0] : k.A[ia][1], Bk = neg ? k.B[ia][0] : k.B[ia][1], Ck = neg ? k.Cc[ia][0] : k.Cc[ia][1];
        K = Ak + Bk * arg + Ck * (arg * arg);
    } else {
        double sn, cs;
        const double x1 = (1.0 - arg) * (1.0 + arg);
        lode_sin_cos(arg, x1 * mc_rsqrt(x1), sn, cs);
        K = cs - k.k_lin[ia] * sn;
    }
    return I1 / 3.0 * k.sin_a[ia] + sqrt(J2 * (K * K) + k.ag2s2[ia]) - k.c * k.cos_a[ia];
}

// Elastic branch (:424-425, :442-443): r = [sigma - sigma_n - C deps, dlambda], J = I, one iteration.
// res0 = -(C deps); deps == 0 gives norm_res0 == 0 and 0/0 = NaN > tol is false: ZERO iterations, sigma =
// sigma_n, C_tang = 0 — the reference's behaviour (:500-505), kept.
DXO_HD void elastic_point(const Const& k, const double* sn, const double* Ce, const double* trial, Result& R) {
    const double n0 = sqrt(dot4(Ce, Ce));
    R.dlambda = 0.0;
    if (!(n0 / n0 > k.tol)) {
        for (int i = 0; i < 4; ++i) R.sigma[i] = sn[i];
        for (int i = 0; i < 16; ++i) R.C_tang[i] = 0.0;
        R.niter = 0;
        R.norm_res = n0;
        return;
    }
    double acc = 0.0;
    for (int i = 0; i < 4; ++i) {
        R.sigma[i] = trial[i];
        const double ri = trial[i] - sn[i] - Ce[i];
        acc += ri * ri;
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) R.C_tang[i * 4 + j] = ((i < 3 && j < 3) ? k.lmbda : 0.0) + (i == j ? k.mu2 : 0.0);
    R.niter = 1;
    R.norm_res = sqrt(acc);
}

// State of one plastic point between Newton passes: y = (sig, dl), Y = d y / d deps (5 x 4).
// WHERE the bulky, rarely touched part lives — the inputs (deps, sn: read once per pass, by the residual) and Y (touched
// one column at a time at the end of a pass) — is a policy: LaneRegs keeps them in the struct (registers on the GPU;
// the CPU build and mc_point), the Newton kernel of mohr_coulomb.hip parks them in the wave's LDS slice so that the
// 28 doubles = 56 VGPRs are not live across the surface / Hessian / third-derivative arithmetic of a pass. The
// arithmetic and its order are the same for every policy.
struct LaneRegs {
    double deps[4], sn[4];
    double Y[5][4];
    DXO_HD void set_inputs(const double* d, const double* s) {
        for (int i = 0; i < 4; ++i) { deps[i] = d[i]; sn[i] = s[i]; }
    }
    DXO_HD void get_inputs(double* d, double* s) const {
        for (int i = 0; i < 4; ++i) { d[i] = deps[i]; s[i] = sn[i]; }
    }
    DXO_HD void get_col(int m, double* v5) const {
        for (int i = 0; i < 5; ++i) v5[i] = Y[i][m];
    }
    DXO_HD void set_col(int m, const double* v5) {
        for (int i = 0; i < 5; ++i) Y[i][m] = v5[i];
    }
};

template <class Store>
struct LaneT {
    Store st;
    double sig[4], dl;
    double norm0, norm;   // norm_res0 (:501) and the current residual norm
    int32_t niter;
};
using Lane = LaneT<LaneRegs>;

template <class Store>
DXO_HD void lane_init(LaneT<Store>& L, const double* deps, const double* sn) {
    L.st.set_inputs(deps, sn);
    for (int i = 0; i < 4; ++i) L.sig[i] = sn[i];   // :496-498
    L.dl = 0.0;
    const double zero[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < 4; ++m) L.st.set_col(m, zero);
    L.norm0 = -1.0;  // "not evaluated yet"
    L.norm = 0.0;
    L.niter = 0;
}

// One pass = evaluate the surface at the current iterate, form r and its norm, test cond_fun (:503-505);
// if the loop continues, do body_fun (:507-522): Newton step + tangent recursion. Returns true when the
// point is finished (converged, NaN, or niter == nitermax); L then holds the reference's outputs.
template <bool SAME, class Store>
DXO_HD bool lane_pass(const Const& k, LaneT<Store>& L) {
    Surf e;
    surf_eval<SAME>(k, L.sig, e);
    double gradg[4], r_sig[4], r_f;
    grad_surface(k, e, 1, gradg);
    {
        double deps[4], sn[4];
        L.st.get_inputs(deps, sn);
        L.norm = residual(k, e, L.sig, L.dl, deps, sn, gradg, r_sig, &r_f);   // :500 / :516-517
    }
    if (L.norm0 < 0.0 || L.norm0 != L.norm0) L.norm0 = (L.niter == 0) ? L.norm : L.norm0;  // :501
    if (!((L.norm / L.norm0 > k.tol) && (L.niter < k.nitermax))) return true;
    // hess(g) as a dense symmetric matrix, once per pass; M = S + dlambda H_g (S = C^-1: (1/E) [[1, -nu, -nu, 0], ..., 1 + nu])
    Sym4 H;
    double av[4], bv[4];
    hess_dense(e, 1, H, av, bv);
    Sym4 M;
    {
        const double sd = k.inv_E * ((1.0 + k.nu) - k.nu), so = -(k.inv_E * k.nu);
        M.m[0] = sd + L.dl * H.m[0]; M.m[1] = so + L.dl * H.m[1]; M.m[2] = so + L.dl * H.m[2]; M.m[3] = L.dl * H.m[3];
        M.m[4] = sd + L.dl * H.m[4]; M.m[5] = so + L.dl * H.m[5]; M.m[6] = L.dl * H.m[6];
        M.m[7] = sd + L.dl * H.m[7]; M.m[8] = L.dl * H.m[8];
        M.m[9] = k.inv_E * (1.0 + k.nu) + L.dl * H.m[9];
    }
    Ldl F;
    ldl_factor(M, F);
    double gradf[4];
    if constexpr (SAME) { for (int i = 0; i < 4; ++i) gradf[i] = gradg[i]; }
    else grad_surface(k, e, 0, gradf);
    // Newton step t = J^-1 r
    double rho[4], xh[4], bh[4];
    S_times(k, r_sig, rho);
    ldl_solve(F, rho, xh);
    ldl_solve(F, gradg, bh);
    const double icb = mc_recip(dot4(gradf, bh));   // the Schur complement's pivot: one division for the step and the four tangent columns
    const double t_l = (dot4(gradf, xh) - r_f) * icb;
    double t_s[4];
    for (int i = 0; i < 4; ++i) t_s[i] = xh[i] - bh[i] * t_l;
    // tangent recursion Y <- J^-1 ([C; 0] + (D J[Y]) t), column by column (column m of the new Y depends on column m
    // of the old one only, so the update is done in place)
    double Ht[4], Hft[4];
    sym_apply(H, t_s, Ht);
    if constexpr (SAME) { for (int i = 0; i < 4; ++i) Hft[i] = Ht[i]; }
    else hess_apply(e, 0, t_s, Hft);
    Sym4 T;   // T(t_s) = D_t hess(g), dense, once per pass
    third_dense(e, 1, t_s, av, bv, T);
    for (int m = 0; m < 4; ++m) {
        double col[5];
        L.st.get_col(m, col);
        const double v[4] = {col[0], col[1], col[2], col[3]};
        const double dlm = col[4];
        double Tv[4], Hv[4], rhs[4], zh[4];
        sym_apply(T, v, Tv);
        sym_apply(H, v, Hv);
        for (int i = 0; i < 4; ++i) rhs[i] = (i == m ? 1.0 : 0.0) + dlm * Ht[i] + L.dl * Tv[i] + t_l * Hv[i];
        const double nu_m = dot4(Hft, v);
        ldl_solve(F, rhs, zh);
        const double mu_m = (dot4(gradf, zh) - nu_m) * icb;
        for (int i = 0; i < 4; ++i) col[i] = zh[i] - bh[i] * mu_m;
        col[4] = mu_m;
        L.st.set_col(m, col);
    }
    for (int i = 0; i < 4; ++i) L.sig[i] -= t_s[i];   // y <- y + solve(j, -r), :513-514
    L.dl -= t_l;
    L.niter += 1;                                      // :520
    return false;
}

DXO_HD void return_map(const Const& k, const double* deps, const double* sn, Result& R) {
    double Ce[4], trial[4];
    C_times(k, deps, Ce);
    for (int i = 0; i < 4; ++i) trial[i] = sn[i] + Ce[i];
    R.yielding = f_value(k, trial);                                  // :422, :531
    if (R.yielding <= 0.0) {
        elastic_point(k, sn, Ce, trial, R);
        return;
    }
    Lane L;
    lane_init(L, deps, sn);
    if (k.same_angle) { while (!lane_pass<true>(k, L)) {} }
    else { while (!lane_pass<false>(k, L)) {} }
    for (int i = 0; i < 4; ++i) R.sigma[i] = L.sig[i];
    for (int j = 0; j < 4; ++j) {
        double col[5];
        L.st.get_col(j, col);
        for (int i = 0; i < 4; ++i) R.C_tang[i * 4 + j] = col[i];
    }
    R.niter = L.niter;
    R.norm_res = L.norm;
    R.dlambda = L.dl;
}

// Host-side constant folding of the model parameters (dxo_mc_params field order).
inline Const make_const(double E, double nu, double c, double phi, double psi, double theta_T, double a, double tol,
                        int32_t nitermax) {
    Const k;
    k.lmbda = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));   // :405
    k.mu2 = 2.0 * (E / (2.0 * (1.0 + nu)));                // :406
    k.inv_E = 1.0 / E;
    k.nu = nu;
    k.c = c;
    k.theta_T = theta_T;
    k.sin3T = sin(3.0 * theta_T);
    k.tol = tol;
    k.nitermax = nitermax;
    k.same_angle = (phi == psi) ? 1 : 0;
    const double ang[2] = {phi, psi};
    const double coeff3 = 18.0 * cos(3.0 * theta_T) * cos(3.0 * theta_T) * cos(3.0 * theta_T);  // :310
    for (int ia = 0; ia < 2; ++ia) {
        const double al = ang[ia];
        k.sin_a[ia] = sin(al);
        k.cos_a[ia] = cos(al);
        k.k_lin[ia] = (1.0 / sqrt(3.0)) * sin(al);
        const double ag = a * tan(phi) / tan(al);           // :348-349
        k.ag2s2[ia] = ag * ag * sin(al) * sin(al);
        const double coeff1 = cos(theta_T) - (1.0 / sqrt(3.0)) * sin(al) * sin(theta_T);  // :302-303
        for (int sg = 0; sg < 2; ++sg) {
            const double sign = sg == 0 ? -1.0 : 1.0;
            const double coeff2 = sign * sin(theta_T) + (1.0 / sqrt(3.0)) * sin(al) * cos(theta_T);  // :306-307
            const double Cc = (-cos(3.0 * theta_T) * coeff1 - 3.0 * sign * sin(3.0 * theta_T) * coeff2) / coeff3;  // :313-316
            const double Bc = (sign * sin(6.0 * theta_T) * coeff1 - 6.0 * cos(6.0 * theta_T) * coeff2) / coeff3;   // :319-322
            const double Ac = -(1.0 / sqrt(3.0)) * sin(al) * sign * sin(theta_T) - Bc * sign * sin(3 * theta_T) -
                              Cc * sin(3.0 * theta_T) * sin(3.0 * theta_T) + cos(theta_T);                          // :325-331
            k.A[ia][sg] = Ac;
            k.B[ia][sg] = Bc;
            k.Cc[ia][sg] = Cc;
        }
    }
    return k;
}

}  // namespace mc
