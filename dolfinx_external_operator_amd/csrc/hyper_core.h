// hyper_core.h — per-point fp64 arithmetic shared by the hyperelastic kernels (icnn.hip, field_ops.hip): the chain rule from
// the feature space (K1, K2, K3) to F, and the analytic Isihara model on top of it.
// Reference: doc/demo/demo_hyperelasticity.py — features :263-283, P = grad_F W + F @ H :433-439, jacfwd(P) :448,
// H correction :371-380, analytic energy :686-703.
#pragma once

#include "dxo_common.h"

namespace {

// x^(-2/3), x > 0, for the features: an fp32 seed (v_log_f32 / v_exp_f32) and two Newton steps on y^-3 = x^2 in fp64 (the error
// squares twice: 1e-6 -> 1e-12 -> fp64 rounding, within 2 ulp) instead of the library pow (about 120 fp64 instructions, a table of
// constants and a long dependent chain). x = 0, inf, nan return the seed's inf / 0 / nan as pow does.
__device__ __forceinline__ double hyper_pow_m23(double x) {
    const double c = x * x;
    const float y0 = __builtin_amdgcn_exp2f(-0.6666666666666666f * __builtin_amdgcn_logf((float)x));
    double y = (double)y0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double r = fma(-(c * (y * y)), y, 1.0);   // 1 - x^2 y^3
        y = fma(y * (1.0 / 3.0), r, y);
    }
    return y == y ? y : (double)y0;
}

// Chain rule from the network input x = (K1, K2, K3) to F in fp64 (shared by both kernels).
// dK_k = kt_k gt + kD_k gD with gt = grad |F|^2 = 2F, gD = grad det F = cof F; W = grad_x y, hx = hess_x y packed
// (00, 01, 02, 11, 12, 22). Writes P = grad_F W_NN + F @ H (:433-439) and dP[i][j] = dP_i/dF_j.
template <typename T>
__device__ __forceinline__ void icnn_chain(const double (&Fv)[4], const double (&kt)[3], const double (&kD)[3],
                                           const double (&ktD)[3], const double (&kDD)[3], const T* y1, const T* hx,
                                           const double* Hc, double* __restrict__ dPp, double* __restrict__ Pp) {
    const double Wk[3] = {(double)y1[0], (double)y1[1], (double)y1[2]};
    const double Hk[3][3] = {{(double)hx[0], (double)hx[1], (double)hx[2]},
                             {(double)hx[1], (double)hx[3], (double)hx[4]},
                             {(double)hx[2], (double)hx[4], (double)hx[5]}};
    double ca = 0, cb = 0, cc = 0, cd = 0, ett = 0, etD = 0, eDD = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ca += Wk[k] * kt[k]; cb += Wk[k] * kD[k]; cc += Wk[k] * ktD[k]; cd += Wk[k] * kDD[k];
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            ett += Hk[k][l] * kt[k] * kt[l];
            etD += Hk[k][l] * kt[k] * kD[l];
            eDD += Hk[k][l] * kD[k] * kD[l];
        }
    }
    const double gt[4] = {2.0 * Fv[0], 2.0 * Fv[1], 2.0 * Fv[2], 2.0 * Fv[3]};
    const double gD[4] = {Fv[3], -Fv[2], -Fv[1], Fv[0]};
    // H = [[h0,h1,0,0],[h2,h3,0,0],[0,0,h0,h1],[0,0,h2,h3]] (:371-380)
    const double FH[4] = {Fv[0] * Hc[0] + Fv[1] * Hc[2], Fv[0] * Hc[1] + Fv[1] * Hc[3],
                          Fv[2] * Hc[0] + Fv[3] * Hc[2], Fv[2] * Hc[1] + Fv[3] * Hc[3]};
    double Pv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Pv[i] = ca * gt[i] + cb * gD[i] + FH[i];
    reinterpret_cast<dxo_f64x2*>(Pp)[0] = dxo_f64x2{Pv[0], Pv[1]};
    reinterpret_cast<dxo_f64x2*>(Pp)[1] = dxo_f64x2{Pv[2], Pv[3]};
    const double ctD = cc + etD, cDD = cd + eDD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double rowv[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const double hD = ((i == 0 && jj == 3) || (i == 3 && jj == 0)) ? 1.0 : (((i == 1 && jj == 2) || (i == 2 && jj == 1)) ? -1.0 : 0.0);
            const int bi = i >> 1, bj = jj >> 1;
            const double Hji = (bi == bj) ? Hc[(jj & 1) * 2 + (i & 1)] : 0.0;   // d(F @ H)_i / dF_j = H[j][i]
            rowv[jj] = (i == jj ? 2.0 * ca : 0.0) + cb * hD + ett * gt[i] * gt[jj] + ctD * (gt[i] * gD[jj] + gD[i] * gt[jj]) +
                       cDD * gD[i] * gD[jj] + Hji;
        }
        reinterpret_cast<dxo_f64x2*>(dPp + i * 4)[0] = dxo_f64x2{rowv[0], rowv[1]};
        reinterpret_cast<dxo_f64x2*>(dPp + i * 4)[1] = dxo_f64x2{rowv[2], rowv[3]};
    }
}

// ---- analytic Isihara energy (demo_hyperelasticity.py:686-703): W = c1 (I1bar - 3) + c2 (I2bar - 3) + c3 (I1bar - 3)^2 + c4 (J - 1)^2,
// a quadratic in the SAME features the network sees, so P and dP/dF come out of the chain rule above with
// grad_x W = (c1 + 2 c3 K1, c2, c4), hess_x W = diag(2 c3, 0, 0). UFL's J = det F: for det F <= 0 the real power is NaN.
struct IsiPrm { double c1, c2, c3, c4; };

__device__ __forceinline__ void isihara_point(const IsiPrm& prm, const double (&Fv)[4], double* __restrict__ dPp, double* __restrict__ Pp) {
    const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
    const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
    const double iD = 1.0 / D;
    const double m = D > 0.0 ? hyper_pow_m23(D) : __builtin_nan(""), nn = m * m;
    const double K1 = (t + 1.0) * m - 3.0;
    const double kt[3] = {m, nn, 0.0};
    const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                          2.0 * (D - 1.0)};
    const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
    const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                           -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
    const double y1[3] = {prm.c1 + 2.0 * prm.c3 * K1, prm.c2, prm.c4};
    const double hx[6] = {2.0 * prm.c3, 0.0, 0.0, 0.0, 0.0, 0.0};
    const double Hzero[4] = {0.0, 0.0, 0.0, 0.0};
    icnn_chain<double>(Fv, kt, kD, ktD, kDD, y1, hx, Hzero, dPp, Pp);
}

}  // namespace
