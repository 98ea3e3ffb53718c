/* icnn_oracle_c.c — C restatement of the reference's ICNN hyperelastic operator, point by point with OpenMP over points.
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline legs): never linked into or imported by the
 * product package. It is the per-point form of oracle/icnn_oracle.py (which states the same jets with NumPy einsums) and exists so
 * that the CPU baseline of BASELINE config 5 is a compiled, threaded port rather than an interpreter-bound checker.
 *
 * Reference: doc/demo/demo_hyperelasticity.py
 *   convexLinear :221-239 (softplus of the weights, :238), ICNN.forward :256-300 (features :263-283, fp32 cast :286,
 *   z = L0(x) :289, z = softplus(z softplus(W)^T + skip(x)); z = z^2 / 12 :290-295, output :299),
 *   H correction :362-381, compute_stress_local :429-443 (P = grad_F W_NN + F @ H), jacfwd :448, dP_dF_impl :451-456.
 * The reference differentiates with torch.func; here every neuron carries its value, 3 first and 6 (symmetric) second
 * derivatives with respect to x = (K1, K2, K3), in fp32 like the reference's network; features and chain rule in fp64.
 * Parity: pinned through tests/golden/icnn_isihara.npz (the reference's own classes executed under torch) and against
 * oracle/icnn_oracle.py (tests/test_oracle_golden.py), at the fp32 noise level of a different summation order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

#define NH 64

typedef struct {
    const float* W0;  /* [64][3]  layers.0.weight */
    const float* b0;  /* [64]     layers.0.bias   */
    const float* W1;  /* [64][64] layers.1.weights (raw: softplus applied here, :238) [out][in] */
    const float* S1;  /* [64][3]  skip_layers.1.weight */
    const float* c1;  /* [64]     skip_layers.1.bias   */
    const float* W2;
    const float* S2;
    const float* c2;
    const float* W3;  /* [1][64] */
    const float* S3;  /* [1][3]  */
} oracle_icnn_weights;

static float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); } /* torch softplus, beta 1, threshold 20 */

typedef struct { float v, d[3], h[6]; } jet;   /* h: xx xy xz yy yz zz */

/* one hidden layer: a = z Wp + S x + c; z <- softplus(a)^2 / 12 with first and second derivatives (:290-295) */
static void layer(const jet* zin, const float* Wp /* [in][out] softplus'd */, const float* S, const float* c, const float* x, jet* zout) {
    for (int o = 0; o < NH; ++o) {
        float a = c[o] + S[o * 3 + 0] * x[0] + S[o * 3 + 1] * x[1] + S[o * 3 + 2] * x[2];
        float da[3] = {S[o * 3 + 0], S[o * 3 + 1], S[o * 3 + 2]};
        float ha[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < NH; ++i) {
            const float w = Wp[i * NH + o];
            a += zin[i].v * w;
            for (int k = 0; k < 3; ++k) da[k] += zin[i].d[k] * w;
            for (int k = 0; k < 6; ++k) ha[k] += zin[i].h[k] * w;
        }
        float sp, s1, s2;
        if (a > 20.0f) { sp = a; s1 = 1.0f; s2 = 0.0f; }
        else { sp = softplus_f(a); s1 = 1.0f / (1.0f + expf(-a)); s2 = s1 * (1.0f - s1); }
        const float p1 = sp * s1 / 6.0f, p2 = (s1 * s1 + sp * s2) / 6.0f;
        zout[o].v = sp * sp / 12.0f;
        for (int k = 0; k < 3; ++k) zout[o].d[k] = p1 * da[k];
        static const int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};
        for (int k = 0; k < 6; ++k) zout[o].h[k] = p1 * ha[k] + p2 * da[A[k]] * da[B[k]];
    }
}

/* y, dy/dx, d2y/dx2 (symmetric 3x3 as 6) of the network at x (fp32) */
static void network(const oracle_icnn_weights* w, const float* Wp1, const float* Wp2, const float* w3p, const float* s3p,
                    const float* x, float* dy, float* hy) {
    jet z0[NH], z1[NH], z2[NH];
    for (int o = 0; o < NH; ++o) {
        z0[o].v = w->b0[o] + w->W0[o * 3] * x[0] + w->W0[o * 3 + 1] * x[1] + w->W0[o * 3 + 2] * x[2];   /* :289 */
        for (int k = 0; k < 3; ++k) z0[o].d[k] = w->W0[o * 3 + k];
        for (int k = 0; k < 6; ++k) z0[o].h[k] = 0.0f;
    }
    layer(z0, Wp1, w->S1, w->c1, x, z1);
    layer(z1, Wp2, w->S2, w->c2, x, z2);
    for (int k = 0; k < 3; ++k) dy[k] = s3p[k];
    for (int k = 0; k < 6; ++k) hy[k] = 0.0f;
    for (int i = 0; i < NH; ++i) {                                   /* :299 */
        for (int k = 0; k < 3; ++k) dy[k] += z2[i].d[k] * w3p[i];
        for (int k = 0; k < 6; ++k) hy[k] += z2[i].h[k] * w3p[i];
    }
}

/* features x = (K1, K2, K3) of F and their first / second derivatives w.r.t. F (fp64; :263-283) */
static void features(const double* F, double* K, double dK[3][4], double d2K[3][4][4]) {
    const double t = F[0] * F[0] + F[1] * F[1] + F[2] * F[2] + F[3] * F[3];
    const double D = F[0] * F[3] - F[1] * F[2];
    const double aD = fabs(D), sg = D < 0.0 ? -1.0 : (D > 0.0 ? 1.0 : 0.0);
    const double m = pow(aD, -2.0 / 3.0), nn = m * m;
    K[0] = (t + 1.0) * m - 3.0; K[1] = (t + D * D) * nn - 3.0; K[2] = (aD - 1.0) * (aD - 1.0);
    const double kt[3] = {m, nn, 0.0};
    const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m / D, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn / D, 2.0 * (aD - 1.0) * sg};
    const double ktD[3] = {(-2.0 / 3.0) * m / D, (-4.0 / 3.0) * nn / D, 0.0};
    const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m / (D * D), -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn / (D * D), 2.0};
    const double gt[4] = {2.0 * F[0], 2.0 * F[1], 2.0 * F[2], 2.0 * F[3]};
    const double gD[4] = {F[3], -F[2], -F[1], F[0]};
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 4; ++i) {
            dK[k][i] = kt[k] * gt[i] + kD[k] * gD[i];
            for (int j = 0; j < 4; ++j) {
                const double Ht = i == j ? 2.0 : 0.0;
                const double HD = (i + j == 3) ? ((i == 0 || i == 3) ? 1.0 : -1.0) : 0.0;
                d2K[k][i][j] = kt[k] * Ht + kD[k] * HD + ktD[k] * (gt[i] * gD[j] + gD[i] * gt[j]) + kDD[k] * gD[i] * gD[j];
            }
        }
}

/* dP (N,4,4) with dP[i][j] = dP_i / dF_j, P (N,4); H_out (16 floats, row-major 4x4) = the correction of :362-381 */
int oracle_icnn(const oracle_icnn_weights* w, int64_t n, const double* F, double* dP, double* P, float* H_out, int nthreads) {
    if (!w || n < 0 || (n > 0 && (!F || !dP || !P))) return -1;
    float* Wp1 = (float*)malloc(sizeof(float) * NH * NH);
    float* Wp2 = (float*)malloc(sizeof(float) * NH * NH);
    float w3p[NH], s3p[3];
    for (int o = 0; o < NH; ++o)
        for (int i = 0; i < NH; ++i) {     /* softplus(W)^T: [in][out] */
            Wp1[i * NH + o] = softplus_f(w->W1[o * NH + i]);
            Wp2[i * NH + o] = softplus_f(w->W2[o * NH + i]);
        }
    for (int i = 0; i < NH; ++i) w3p[i] = softplus_f(w->W3[i]);
    for (int k = 0; k < 3; ++k) s3p[k] = softplus_f(w->S3[k]);
    /* H = -P_NN(F = I), evaluated like the reference with an fp32 F_0 (:371-381) */
    float h[4];
    {
        const double F0[4] = {1.0, 0.0, 0.0, 1.0};
        double K[3], dK[3][4], d2K[3][4][4];
        features(F0, K, dK, d2K);
        const float x[3] = {(float)K[0], (float)K[1], (float)K[2]};
        float dy[3], hy[6];
        network(w, Wp1, Wp2, w3p, s3p, x, dy, hy);
        for (int i = 0; i < 4; ++i) h[i] = -(float)((double)dy[0] * (float)dK[0][i] + (double)dy[1] * (float)dK[1][i] + (double)dy[2] * (float)dK[2][i]);
    }
    const float Hm[16] = {h[0], h[1], 0, 0, h[2], h[3], 0, 0, 0, 0, h[0], h[1], 0, 0, h[2], h[3]};
    if (H_out) for (int i = 0; i < 16; ++i) H_out[i] = Hm[i];
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const double* Fp = F + p * 4;
        double K[3], dK[3][4], d2K[3][4][4];
        features(Fp, K, dK, d2K);
        const float x[3] = {(float)K[0], (float)K[1], (float)K[2]};       /* :286 */
        float dyf[3], hyf[6];
        network(w, Wp1, Wp2, w3p, s3p, x, dyf, hyf);
        const double dy[3] = {dyf[0], dyf[1], dyf[2]};
        const double hy[3][3] = {{hyf[0], hyf[1], hyf[2]}, {hyf[1], hyf[3], hyf[4]}, {hyf[2], hyf[4], hyf[5]}};
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += dy[k] * dK[k][i];
            for (int r = 0; r < 4; ++r) s += Fp[r] * (double)Hm[r * 4 + i];          /* F @ H, :439 */
            P[p * 4 + i] = s;
            for (int j = 0; j < 4; ++j) {
                double t = (double)Hm[j * 4 + i];                                   /* d(F @ H)_i / dF_j = H[j][i] */
                for (int k = 0; k < 3; ++k) {
                    t += dy[k] * d2K[k][i][j];
                    for (int l = 0; l < 3; ++l) t += hy[k][l] * dK[k][i] * dK[l][j];
                }
                dP[(p * 4 + i) * 4 + j] = t;
            }
        }
    }
    free(Wp1);
    free(Wp2);
    return 0;
}

/* Analytic Isihara model (doc/demo/demo_hyperelasticity.py:686-703, a UFL form in the reference): W = c1 K1 + c2 K2 + c3 K1^2 + c4 K3 with
 * the (K1, K2, K3) of `features`; P = dW/dF, dP = d2W/dF2 by the chain rule — the per-point form of oracle/icnn_oracle.py::isihara_stress_tangent
 * (:122-141), OpenMP over points. det F <= 0: NaN, as there. */
int oracle_isihara(const double* c, int64_t n, const double* F, double* dP, double* P, int nthreads) {
    if (!c || n < 0 || (n > 0 && (!F || !dP || !P))) return -1;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        const double* Fp = F + p * 4;
        if (Fp[0] * Fp[3] - Fp[1] * Fp[2] <= 0.0) {
            for (int i = 0; i < 4; ++i) P[p * 4 + i] = NAN;
            for (int i = 0; i < 16; ++i) dP[p * 16 + i] = NAN;
            continue;
        }
        double K[3], dK[3][4], d2K[3][4][4];
        features(Fp, K, dK, d2K);
        const double dy[3] = {c[0] + 2.0 * c[2] * K[0], c[1], c[3]};
        for (int i = 0; i < 4; ++i) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += dy[k] * dK[k][i];
            P[p * 4 + i] = s;
            for (int j = 0; j < 4; ++j) {
                double t = 2.0 * c[2] * dK[0][i] * dK[0][j];
                for (int k = 0; k < 3; ++k) t += dy[k] * d2K[k][i][j];
                dP[(p * 4 + i) * 4 + j] = t;
            }
        }
    }
    return 0;
}
