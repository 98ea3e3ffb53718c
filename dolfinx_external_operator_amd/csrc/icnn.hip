// icnn.hip — hyperelastic stress P = dW/dF and tangent dP/dF of the input-convex neural network (ICNN)
// surrogate of the reference's hyperelasticity demo, one lane per quadrature point.
//
// Reference: doc/demo/demo_hyperelasticity.py — convexLinear :221-239, ICNN.forward :256-300,
// H correction :362-381, compute_stress_local :429-443, vmap(jacfwd(.)) :448, dP_dF_impl :451-456.
// The reference evaluates the network in fp32 (`.float()`, :286) and differentiates it with torch.func
// (reverse mode for P, forward-over-reverse for the tangent). Here the derivatives are closed form:
//
//   features (fp64)   x = (K1, K2, K3)(F) depend on F only through t = |F|^2 and D = det F, so
//                     dx/dF and d2x/dF2 are rank-structured in grad t = 2F and grad D = cof F.
//   network (fp32 by default, fp64 for the tolerance study)
//                     layer 0 is linear with NO activation (:289), so layer 1's pre-activation is affine in x:
//                     a1 = A1 x + d1 with A1 = softplus(W1) W0 + S1 folded once on the host. Then
//                       h1 = phi(a1), a2 = W2p h1 + S2 x + c2, y = w3p . phi(a2) + s3p . x,  phi(a) = softplus(a)^2/12
//                       grad_x y = sum_j delta_j g_j + s3p,    g_j = sum_i W2p_ji phi'(a1_i) A1_i + S2_j,  delta_j = w3p_j phi'(a2_j)
//                       hess_x y = sum_j w3p_j phi''(a2_j) g_j g_j^T + sum_i beta_i phi''(a1_i) A1_i A1_i^T,  beta = W2p^T delta
//                     i.e. three mat-vec shaped products with the 64x64 matrix W2p per point: ~20.5 k FMA.
//
// Roofline: compute (fp32 VALU). 192 B/point of HBM traffic against ~45 kflop/point: > 200 flop/B.
// Weights are wave-uniform: they are read through the scalar path (s_load into SGPRs) and enter the FMAs
// as scalar operands; the per-point vectors h1, phi'(a1), beta live in VGPRs (3 x 64 floats per lane).
// The 64x64 products are a genuine dense contraction (SURVEY.md 8d notes MFMA is legitimate here);
// gfx950's fp32-input MFMA runs at the fp32 VALU rate, so round 1 keeps the VALU form.
#include <cmath>

#include "dxo_common.h"

namespace {

constexpr int NH = 64;

template <typename T>
struct IcnnDev {
    const T* A1;    // [64][4]: A1[i][0..2], d1[i]
    const T* W2B;   // [64][4][64]: row j = { W2p[j][:], W2p[j][:] * A1[:][0], * A1[:][1], * A1[:][2] } (four 64-vectors,
                    // each fetched with s_load_dwordx16 bursts)
    const T* W2;    // [64][64]: W2p[j][i] (dense, the MFMA kernel's operand source)
    const T* S2;    // [64][4]: S2[j][0..2], c2[j]
    const T* w3;    // [64]: w3p
    T s3[3];        // s3p
    double H[4];    // H_flat (:371)
};

struct dxo_icnn_impl {
    void* dev = nullptr;  // one slab holding both precisions
    IcnnDev<float> f32;
    IcnnDev<double> f64;
};

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log1p(T x);
template <> __device__ __forceinline__ float t_log1p<float>(float x) { return log1pf(x); }
template <> __device__ __forceinline__ double t_log1p<double>(double x) { return log1p(x); }

// softplus (beta = 1, threshold = 20 as torch.nn.functional.softplus) with first and second derivative
// fp32 softplus on the hardware transcendental unit (v_exp_f32 / v_log_f32, ~1 ulp): ~10 instructions instead of
// ~120 for expf + log1pf. log(1 + e) loses RELATIVE accuracy for e < 1e-4, but there softplus < 1e-4 and only
// enters through softplus^2/12 — absolute error < 1e-12 of the network output.
__device__ __forceinline__ void softplus3_fast(float a, float& sp, float& s1, float& s2) {
    // raw v_exp_f32 / v_log_f32 / v_rcp_f32: the arguments are bounded (e <= e^20, 1 + e >= 1), so the
    // denormal-range fix-ups that __expf / __logf add are dead weight here
    const float e = __builtin_amdgcn_exp2f(fminf(a, 20.0f) * 1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const bool big = a > 20.0f;
    sp = big ? a : __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    s1 = big ? 1.0f : e * r;
    s2 = big ? 0.0f : e * r * r;
}

template <typename T>
__device__ __forceinline__ void softplus3(T a, T& sp, T& s1, T& s2) {
    if (a > T(20)) {
        sp = a; s1 = T(1); s2 = T(0);
    } else {
        const T e = t_exp<T>(a);
        const T r = T(1) / (T(1) + e);
        sp = t_log1p<T>(e);
        s1 = e * r;
        s2 = s1 * r;
    }
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

template <typename T>
struct IcnnSmall {  // by-value kernel argument: lands in SGPRs
    T s3[3];
    double H[4];
};

// The weight arrays are separate `const T* __restrict__` kernel arguments on purpose: only then can the
// compiler prove they are never written and fetch them through the scalar path (s_load -> SGPR operands).
// Passed as pointers inside a struct they were loaded with per-lane global_load_dwordx4 (64x redundant,
// 256 VGPRs for one weight row, h1/u spilled to scratch).
template <typename T>
__global__ __launch_bounds__(DXO_BLOCK) void icnn_point(const T* __restrict__ wA1, const T* __restrict__ wW2B,
                                                        const T* __restrict__ wS2, const T* __restrict__ ww3,
                                                        IcnnSmall<T> small, int64_t n, const double* __restrict__ F,
                                                        double* __restrict__ dP, double* __restrict__ P) {
    struct { const T* A1; const T* W2B; const T* S2; const T* w3; const T* s3; const double* H; } w =
        {wA1, wW2B, wS2, ww3, small.s3, small.H};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const dxo_f64x2 f01 = reinterpret_cast<const dxo_f64x2*>(F + p * 4)[0];
        const dxo_f64x2 f23 = reinterpret_cast<const dxo_f64x2*>(F + p * 4)[1];
        const double Fv[4] = {f01.x, f01.y, f23.x, f23.y};
        // ---- features and their (t, D) partials, fp64 (:263-283)
        const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
        const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
        const double aD = fabs(D), sg = D < 0.0 ? -1.0 : 1.0, iD = 1.0 / D;
        const double m = pow(aD, -2.0 / 3.0), nn = m * m;
        const double K[3] = {(t + 1.0) * m - 3.0, (t + D * D) * nn - 3.0, (aD - 1.0) * (aD - 1.0)};
        const double kt[3] = {m, nn, 0.0};
        const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                              2.0 * (aD - 1.0) * sg};
        const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
        const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                               -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
        const T x0 = (T)K[0], x1 = (T)K[1], x2 = (T)K[2];   // the `.float()` of :286 when T = float
        // ---- layer 1 (layer 0 folded in): h1 = phi(a1), u = phi'(a1)
        T h1[NH], u[NH], beta[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const T a = w.A1[i * 4 + 0] * x0 + w.A1[i * 4 + 1] * x1 + w.A1[i * 4 + 2] * x2 + w.A1[i * 4 + 3];
            T sp, s1, s2;
            softplus3<T>(a, sp, s1, s2);
            h1[i] = sp * sp * T(1.0 / 12.0);      // :293-294
            u[i] = sp * s1 * T(1.0 / 6.0);
            beta[i] = T(0);
            // Pin both in registers: without this the compiler SINKS the whole softplus chain into the
            // 64-iteration neuron loop below (it can recompute h1[i] from x) and evaluates 64 x 64 exp/log1p
            // per point instead of 64 — measured 14x slower.
            asm volatile("" : "+v"(h1[i]), "+v"(u[i]));
        }
        // ---- layer 2 neuron by neuron: a2_j, g_j = grad_x a2_j, then the contributions of neuron j
        T y1[3] = {w.s3[0], w.s3[1], w.s3[2]};              // grad_x y
        T hx[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};      // hess_x y: 00, 01, 02, 11, 12, 22
#pragma unroll 1
        for (int j = 0; j < NH; ++j) {
            const T* row = w.W2B + (size_t)j * NH * 4;
            T a2 = w.S2[j * 4 + 0] * x0 + w.S2[j * 4 + 1] * x1 + w.S2[j * 4 + 2] * x2 + w.S2[j * 4 + 3];
            T g0 = w.S2[j * 4 + 0], g1 = w.S2[j * 4 + 1], g2 = w.S2[j * 4 + 2];
#pragma unroll
            for (int i = 0; i < NH; ++i) a2 += row[i] * h1[i];
#pragma unroll
            for (int i = 0; i < NH; ++i) g0 += row[NH + i] * u[i];
#pragma unroll
            for (int i = 0; i < NH; ++i) g1 += row[2 * NH + i] * u[i];
#pragma unroll
            for (int i = 0; i < NH; ++i) g2 += row[3 * NH + i] * u[i];
            T sp, s1, s2;
            softplus3<T>(a2, sp, s1, s2);
            const T w3 = w.w3[j];
            const T delta = w3 * sp * s1 * T(1.0 / 6.0);                    // w3p_j phi'(a2_j)
            const T curv = w3 * (s1 * s1 + sp * s2) * T(1.0 / 6.0);         // w3p_j phi''(a2_j)
            y1[0] += delta * g0; y1[1] += delta * g1; y1[2] += delta * g2;
            hx[0] += curv * g0 * g0; hx[1] += curv * g0 * g1; hx[2] += curv * g0 * g2;
            hx[3] += curv * g1 * g1; hx[4] += curv * g1 * g2; hx[5] += curv * g2 * g2;
#pragma unroll
            for (int i = 0; i < NH; ++i) beta[i] += delta * row[i];
        }
        // ---- second Hessian term: sum_i beta_i phi''(a1_i) A1_i A1_i^T
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const T A0 = w.A1[i * 4 + 0], A1v = w.A1[i * 4 + 1], A2 = w.A1[i * 4 + 2];
            const T a = A0 * x0 + A1v * x1 + A2 * x2 + w.A1[i * 4 + 3];
            T sp, s1, s2;
            softplus3<T>(a, sp, s1, s2);
            const T c = beta[i] * (s1 * s1 + sp * s2) * T(1.0 / 6.0);
            hx[0] += c * A0 * A0; hx[1] += c * A0 * A1v; hx[2] += c * A0 * A2;
            hx[3] += c * A1v * A1v; hx[4] += c * A1v * A2; hx[5] += c * A2 * A2;
        }
        T y1v[3] = {y1[0], y1[1], y1[2]};
        icnn_chain(Fv, kt, kD, ktD, kDD, y1v, hx, w.H, dP + p * 16, P + p * 4);
    }
}

// ------------------------------------------------------------------ fp32 MFMA kernel (variant 1)
// One wave = 64 points. The three mat-vec shaped products with W2p become five 64x64x64 GEMMs per wave on
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise an fmaf chain), everything else runs on the VALU meanwhile:
//   [a2 | g0 | g1 | g2](j, pt) = W2p(j, i) @ [h1 | u A1_0 | u A1_1 | u A1_2](i, pt)      4 GEMMs
//   beta(i, pt)                = W2p^T(i, j) @ delta(j, pt)                                1 GEMM
// Lane l = (h = l >> 5, p = l & 31). Points are processed in two tiles of 32 (pt = 32 t + p).
//   A operand (lane: row = l & 31, k = l >> 5):  A[jt][r]      = W2p[32 jt + p][r + 32 h]        K-step r covers i in {r, r + 32}
//                                                AT[it][jt][q] = W2p[jlo(jt,q) + 4 h][32 it + p]  K-step (jt,q) covers j in {jlo, jlo + 4}
//     — 128 weight registers, loaded ONCE per wave and kept resident across the grid-stride loop.
//   B operand (lane: k = l >> 5, col = l & 31): the lane evaluates layer-1 neuron i = r + 32 h for point 32 t + p
//     itself (softplus on the VALU while the matrix pipe works), so B never has to be shuffled into place.
//   C/D (lane: col = l & 31, row = (q & 3) + 8 (q >> 2) + 4 h): delta is produced in exactly the layout the
//     beta GEMM wants as its B operand (k = h selects row jlo + 4 h), again without a shuffle.
// The only cross-lane traffic is x of the partner half (3 values) and the final 9-value half-wave sums.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }
__device__ __forceinline__ void icnn_lds_fence() {   // wave-private LDS: only the compiler needs ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void icnn_mfma(const float* __restrict__ wA1, const float* __restrict__ wW2,
                                                       const float* __restrict__ wS2, const float* __restrict__ ww3,
                                                       IcnnSmall<float> small, int64_t n, const double* __restrict__ F,
                                                       double* __restrict__ dP, double* __restrict__ P) {
    constexpr int BLOCK = WAVES * 64;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const float4* A1v = reinterpret_cast<const float4*>(wA1);   // A1[i] = (A1_i0, A1_i1, A1_i2, d1_i)
    const float4* S2v = reinterpret_cast<const float4*>(wS2);   // S2[j] = (S2_j0, S2_j1, S2_j2, c2_j)
    // The A operands of all five GEMMs, pre-arranged in MFMA lane order, live in LDS (32 KiB per workgroup,
    // filled once): sA[jt][r][lane], sAT[it][jt][q][lane]. Keeping them in registers instead (128 VGPRs)
    // forced the K loop to be fully unrolled and spilled 2.3 KB per lane.
    __shared__ float sA[2 * 32 * 64];
    __shared__ float sAT[2 * 2 * 16 * 64];
    // per wave: 8 KiB to park 4 accumulator tiles, 8 of their 16 registers at a time (later the 32 beta values),
    // and 8 KiB for delta in C layout = B operand of the beta GEMM. With WAVES = 8 the workgroup uses exactly the
    // CU's 160 KiB and runs 2 waves per SIMD, so one wave's softplus passes overlap the other's MFMAs.
    __shared__ float sStage[WAVES * 32 * 64];
    __shared__ float sDelta[WAVES * 32 * 64];
    float* stage = sStage + wave * (32 * 64);
    float* sdlt = sDelta + wave * (32 * 64);
    for (int e = threadIdx.x; e < 2 * 32 * 64; e += BLOCK) {
        const int l = e & 63, r = (e >> 6) & 31, jt = e >> 11;
        sA[e] = wW2[(32 * jt + (l & 31)) * NH + r + 32 * (l >> 5)];
    }
    for (int e = threadIdx.x; e < 2 * 2 * 16 * 64; e += BLOCK) {
        const int l = e & 63, q = (e >> 6) & 15, jt = (e >> 10) & 1, it = e >> 11;
        sAT[e] = wW2[(32 * jt + (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)) * NH + 32 * it + (l & 31)];
    }
    __syncthreads();

    const int64_t n_tiles = (n + 63) / 64;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += (int64_t)gridDim.x * WAVES) {
        const int64_t pidx = tile * 64 + lane;
        const int64_t pl = pidx < n ? pidx : n - 1;   // tail lanes recompute the last point, never store
        const dxo_f64x2 f01 = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[0];
        const dxo_f64x2 f23 = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[1];
        const double Fv[4] = {f01.x, f01.y, f23.x, f23.y};
        const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
        const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
        const double aD = fabs(D), sg = D < 0.0 ? -1.0 : 1.0, iD = 1.0 / D;
        const double m = pow(aD, -2.0 / 3.0), nn = m * m;
        const float x0 = (float)((t + 1.0) * m - 3.0), x1 = (float)((t + D * D) * nn - 3.0), x2 = (float)((aD - 1.0) * (aD - 1.0));
        const float xp0 = xor32(x0), xp1 = xor32(x1), xp2 = xor32(x2);
        float mine[9];
#pragma unroll 1
        for (int tt = 0; tt < 2; ++tt) {
            const bool own = (tt == h);
            const float xs0 = own ? x0 : xp0, xs1 = own ? x1 : xp1, xs2 = own ? x2 : xp2;
            // Everything below runs in ROLLED loops: the MFMA results are parked in a wave-private LDS slice
            // (lane-linear, each lane re-reads only its own column) purely so that the softplus passes can
            // index them dynamically. Fully unrolled they were scheduled for ILP and spilled > 300 VGPRs.
            float res[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 acc[2][4];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[jt][c][q] = 0.f;
            // the small tables (A1, S2, w3) come from global memory through L1; every rolled loop below fetches the
            // NEXT iteration's row before working on the current one, otherwise each iteration eats an L1 round trip
            float4 a1n = A1v[32 * h];
#pragma unroll 2
            for (int r = 0; r < 32; ++r) {
                const float4 a1 = a1n;
                a1n = A1v[((r + 1) & 31) + 32 * h];
                const float Ar0 = sA[(0 * 32 + r) * 64 + lane], Ar1 = sA[(1 * 32 + r) * 64 + lane];
                const float a = a1.x * xs0 + a1.y * xs1 + a1.z * xs2 + a1.w;
                float sp, s1, s2;
                softplus3_fast(a, sp, s1, s2);
                const float hv = sp * sp * (1.0f / 12.0f);
                const float uv = sp * s1 * (1.0f / 6.0f);
                const float B1 = uv * a1.x, B2 = uv * a1.y, B3 = uv * a1.z;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar0, hv, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar1, hv, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar0, B1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar1, B1, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar0, B2, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar1, B2, acc[1][2], 0, 0, 0);
                acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar0, B3, acc[0][3], 0, 0, 0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ar1, B3, acc[1][3], 0, 0, 0);
            }
#pragma unroll 1
            for (int part = 0; part < 4; ++part) {   // (jt, half): 8 accumulator registers of 4 tiles at a time
                const int jt = part >> 1, q0 = (part & 1) * 8;
#define DXO_PARK(JT, Q0)                                                                  \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                          \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) stage[(c * 8 + q) * 64 + lane] = acc[JT][c][Q0 + q];
                if (part == 0) { DXO_PARK(0, 0) } else if (part == 1) { DXO_PARK(0, 8) } else if (part == 2) { DXO_PARK(1, 0) } else { DXO_PARK(1, 8) }
#undef DXO_PARK
                icnn_lds_fence();
                // ---- neurons of layer 2 (C layout: q -> j): skip connection, softplus, grad / hess contributions
                const int jbase = 32 * jt + 4 * h;
                float4 s2n = S2v[jbase + (q0 & 3) + 8 * (q0 >> 2)];
                float w3n = ww3[jbase + (q0 & 3) + 8 * (q0 >> 2)];
#pragma unroll 2
                for (int qq = 0; qq < 8; ++qq) {
                    const int q = q0 + qq;
                    const float4 s2r = s2n;
                    const float w3 = w3n;
                    {
                        const int qn = q0 + ((qq + 1) & 7);
                        s2n = S2v[jbase + (qn & 3) + 8 * (qn >> 2)];
                        w3n = ww3[jbase + (qn & 3) + 8 * (qn >> 2)];
                    }
                    const float a2 = stage[(0 * 8 + qq) * 64 + lane] + s2r.x * xs0 + s2r.y * xs1 + s2r.z * xs2 + s2r.w;
                    const float g0 = stage[(1 * 8 + qq) * 64 + lane] + s2r.x;
                    const float g1 = stage[(2 * 8 + qq) * 64 + lane] + s2r.y;
                    const float g2 = stage[(3 * 8 + qq) * 64 + lane] + s2r.z;
                    float sp, s1, s2;
                    softplus3_fast(a2, sp, s1, s2);
                    const float delta = w3 * sp * s1 * (1.0f / 6.0f);
                    const float curv = w3 * (s1 * s1 + sp * s2) * (1.0f / 6.0f);
                    res[0] += delta * g0; res[1] += delta * g1; res[2] += delta * g2;
                    res[3] += curv * g0 * g0; res[4] += curv * g0 * g1; res[5] += curv * g0 * g2;
                    res[6] += curv * g1 * g1; res[7] += curv * g1 * g2; res[8] += curv * g2 * g2;
                    sdlt[(jt * 16 + q) * 64 + lane] = delta;   // already the beta GEMM's B operand layout
                }
                icnn_lds_fence();
            }
            // ---- beta = W2p^T delta (K-step jq = (jt, q) covers j in {jlo, jlo + 4})
            f32x16 bacc[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) bacc[it][q] = 0.f;
#pragma unroll 4
            for (int jq = 0; jq < 32; ++jq) {
                const float Bd = sdlt[jq * 64 + lane];
                bacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sAT[(0 * 32 + jq) * 64 + lane], Bd, bacc[0], 0, 0, 0);
                bacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sAT[(1 * 32 + jq) * 64 + lane], Bd, bacc[1], 0, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) stage[(it * 16 + q) * 64 + lane] = bacc[it][q];
            icnn_lds_fence();
            // ---- second Hessian term: sum_i beta_i phi''(a1_i) A1_i A1_i^T (C layout: iq -> i)
            float4 b1n = A1v[4 * h];
#pragma unroll 2
            for (int iq = 0; iq < 32; ++iq) {
                const float4 a1 = b1n;
                {
                    const int in_ = (iq + 1) & 31, qn = in_ & 15;
                    b1n = A1v[32 * (in_ >> 4) + (qn & 3) + 8 * (qn >> 2) + 4 * h];
                }
                const float a = a1.x * xs0 + a1.y * xs1 + a1.z * xs2 + a1.w;
                float sp, s1, s2;
                softplus3_fast(a, sp, s1, s2);
                const float c = stage[iq * 64 + lane] * (s1 * s1 + sp * s2) * (1.0f / 6.0f);
                res[3] += c * a1.x * a1.x; res[4] += c * a1.x * a1.y; res[5] += c * a1.x * a1.z;
                res[6] += c * a1.y * a1.y; res[7] += c * a1.y * a1.z; res[8] += c * a1.z * a1.z;
            }
            icnn_lds_fence();
            // both half-waves hold partial sums of the SAME 32 points: add them, keep the tile this lane owns
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const float tot = res[q] + xor32(res[q]);
                if (own) mine[q] = tot;
            }
        }
        if (pidx < n) {
            const double kt[3] = {m, nn, 0.0};
            const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                                  2.0 * (aD - 1.0) * sg};
            const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
            const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                                   -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
            const float y1f[3] = {mine[0] + small.s3[0], mine[1] + small.s3[1], mine[2] + small.s3[2]};
            icnn_chain(Fv, kt, kD, ktD, kDD, y1f, mine + 3, small.H, dP + pidx * 16, P + pidx * 4);
        }
    }
}

template <typename T>
void launch_icnn(const IcnnDev<T>& m, int blocks, hipStream_t s, int64_t n, const double* F, double* dP, double* P) {
    IcnnSmall<T> small;
    for (int k = 0; k < 3; ++k) small.s3[k] = m.s3[k];
    for (int k = 0; k < 4; ++k) small.H[k] = m.H[k];
    hipLaunchKernelGGL((icnn_point<T>), dim3(blocks), dim3(DXO_BLOCK), 0, s, m.A1, m.W2B, m.S2, m.w3, small, n, F, dP, P);
}

double softplus_host(double x) { return x > 20.0 ? x : std::log1p(std::exp(x)); }

struct IcnnLaunch {
    const dxo_icnn_impl* m;
    int precision;
};

int icnn_launch(dxo_ctx* ctx, const IcnnLaunch& L, int64_t n, const double* F, double* dP, double* P, hipStream_t s) {
    if (n == 0) return DXO_OK;
    int64_t blocks = (n + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (L.precision == 0 && ctx->icnn_variant != 0) {
        // MFMA kernel: one wave = 64 points, weights resident in registers -> persistent grid, 1 wave per SIMD
        IcnnSmall<float> small;
        for (int k = 0; k < 3; ++k) small.s3[k] = L.m->f32.s3[k];
        for (int k = 0; k < 4; ++k) small.H[k] = L.m->f32.H[k];
        // one workgroup per CU: 4 waves (1 per SIMD) or, default, 8 waves (2 per SIMD, all 160 KiB of LDS)
        const int waves = ctx->icnn_variant == 2 ? 4 : 8;
        int64_t mb = (n + waves * 64 - 1) / (waves * 64);
        if (mb > ctx->compute_units) mb = ctx->compute_units;
        if (waves == 8)
            hipLaunchKernelGGL(icnn_mfma<8>, dim3((int)mb), dim3(512), 0, s, L.m->f32.A1, L.m->f32.W2, L.m->f32.S2, L.m->f32.w3,
                               small, n, F, dP, P);
        else
            hipLaunchKernelGGL(icnn_mfma<4>, dim3((int)mb), dim3(256), 0, s, L.m->f32.A1, L.m->f32.W2, L.m->f32.S2, L.m->f32.w3,
                               small, n, F, dP, P);
        return DXO_OK;
    }
    if (L.precision == 0) launch_icnn<float>(L.m->f32, (int)blocks, s, n, F, dP, P);
    else launch_icnn<double>(L.m->f64, (int)blocks, s, n, F, dP, P);
    return DXO_OK;
}

int icnn_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const IcnnLaunch& L = *static_cast<const IcnnLaunch*>(user);
    return icnn_launch(ctx, L, m, (const double*)d_in[0], (double*)d_out[0], (double*)d_out[1], s);
}

// ------------------------------------------------------------------ analytic Isihara energy (demo_hyperelasticity.py:686-703)
// W = c1 (I1bar - 3) + c2 (I2bar - 3) + c3 (I1bar - 3)^2 + c4 (J - 1)^2 with I1bar = J^(-2/3) I1, I2bar = J^(-4/3) I2,
// I1 = tr C + 1, I2 = I1 + J^2 - 1 (plane strain) — i.e. W is a quadratic in the SAME features (K1, K2, K3) the network
// sees, so P = dW/dF and dP/dF come out of the same chain rule with grad_x W = (c1 + 2 c3 K1, c2, c4) and
// hess_x W = diag(2 c3, 0, 0). The reference only has this model as UFL (`P = ufl.diff(W_Isihara, F_)`, :703); as a
// kernel it is the HBM-bound twin of the network operator (192 B per point, a few hundred flop).
// Unlike the network's features (J = sqrt(det C) = |det F|, :279) UFL's J = det F: for det F < 0 the real power
// J^(-2/3) is NaN, and so is the output here.
struct IsiPrm { double c1, c2, c3, c4; };

template <bool NT>
__global__ __launch_bounds__(DXO_BLOCK) void isihara_tile(IsiPrm prm, int64_t n, const double* __restrict__ F,
                                                          double* __restrict__ dP, double* __restrict__ P) {
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * DXO_WAVE * 20];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* Xd = lds + wave * (DXO_WAVE * 20);     // [64][16] tangent rows of this wave's points
    double* Xp = Xd + DXO_WAVE * 16;               // [64][4]  stresses
    const dxo_f64x2* Xd2 = reinterpret_cast<const dxo_f64x2*>(Xd);
    const dxo_f64x2* Xp2 = reinterpret_cast<const dxo_f64x2*>(Xp);
    const double Hzero[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int64_t tile_stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += tile_stride) {
        const int64_t p0 = tile * DXO_WAVE;
        const int npts = (n - p0 < DXO_WAVE) ? (int)(n - p0) : DXO_WAVE;
        dxo_f64x2 f01{1.0, 0.0}, f23{0.0, 1.0};
        if (lane < npts) {
            f01 = reinterpret_cast<const dxo_f64x2*>(F + (p0 + lane) * 4)[0];
            f23 = reinterpret_cast<const dxo_f64x2*>(F + (p0 + lane) * 4)[1];
        }
        const double Fv[4] = {f01.x, f01.y, f23.x, f23.y};
        const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
        const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
        const double iD = 1.0 / D;
        const double m = D > 0.0 ? pow(D, -2.0 / 3.0) : __builtin_nan(""), nn = m * m;
        const double K1 = (t + 1.0) * m - 3.0;
        const double kt[3] = {m, nn, 0.0};
        const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                              2.0 * (D - 1.0)};
        const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
        const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                               -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
        const double y1[3] = {prm.c1 + 2.0 * prm.c3 * K1, prm.c2, prm.c4};
        const double hx[6] = {2.0 * prm.c3, 0.0, 0.0, 0.0, 0.0, 0.0};
        icnn_chain<double>(Fv, kt, kD, ktD, kDD, y1, hx, Hzero, Xd + lane * 16, Xp + lane * 4);
        icnn_lds_fence();
        // output-ordered stores: every store instruction of the wave covers 1 KiB of consecutive addresses
        dxo_f64x2* g_d = reinterpret_cast<dxo_f64x2*>(dP + p0 * 16);
        dxo_f64x2* g_p = reinterpret_cast<dxo_f64x2*>(P + p0 * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < npts * 8) {
                if constexpr (NT) __builtin_nontemporal_store(Xd2[idx], g_d + idx);
                else g_d[idx] = Xd2[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < npts * 2) {
                if constexpr (NT) __builtin_nontemporal_store(Xp2[idx], g_p + idx);
                else g_p[idx] = Xp2[idx];
            }
        }
        icnn_lds_fence();
    }
}

int isihara_launch(dxo_ctx* ctx, const IsiPrm& prm, int64_t n, const double* F, double* dP, double* P, hipStream_t s) {
    if (n == 0) return DXO_OK;
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
    if (ctx->nontemporal != 0) hipLaunchKernelGGL(isihara_tile<true>, dim3(grid), dim3(DXO_BLOCK), 0, s, prm, n, F, dP, P);
    else hipLaunchKernelGGL(isihara_tile<false>, dim3(grid), dim3(DXO_BLOCK), 0, s, prm, n, F, dP, P);
    return DXO_OK;
}

int isihara_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    return isihara_launch(ctx, *static_cast<const IsiPrm*>(user), m, (const double*)d_in[0], (double*)d_out[0],
                          (double*)d_out[1], s);
}

}  // namespace

extern "C" int dxo_isihara(dxo_ctx* ctx, const dxo_isihara_params* prm, int64_t n, int mem, const double* F, double* dP,
                           double* P) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_isihara: params is NULL");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_isihara: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_isihara: bad mem");
    if (n > 0 && (!F || !dP || !P)) return dxo_fail(ctx, DXO_E_NULL, "dxo_isihara: NULL array");
    const uintptr_t all = (uintptr_t)F | (uintptr_t)dP | (uintptr_t)P;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_isihara: arrays must be 8-byte aligned");
    if (mem == DXO_MEM_DEVICE && (all & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_isihara: device arrays must be 16-byte aligned");
    IsiPrm k{prm->c1, prm->c2, prm->c3, prm->c4};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = isihara_launch(ctx, k, n, F, dP, P, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{F, nullptr, 4 * sd}};
    std::vector<dxo_span> out = {{nullptr, dP, 16 * sd}, {nullptr, P, 4 * sd}};
    return dxo_run_host_pipeline(ctx, n, in, out, isihara_chunk, &k);
}

struct dxo_icnn : dxo_icnn_impl {};

extern "C" int dxo_icnn_create(dxo_ctx* ctx, const dxo_icnn_weights* w, dxo_icnn** out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!w || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_create: NULL argument");
    *out = nullptr;
    if (w->n_hidden != NH) return dxo_fail(ctx, DXO_E_DIM, "dxo_icnn_create: n_hidden must be 64 (the reference's [64, 64, 64])");
    if (!w->layers0_weight || !w->layers0_bias || !w->layers1_weights || !w->skip1_weight || !w->skip1_bias ||
        !w->layers2_weights || !w->skip2_weight || !w->skip2_bias || !w->layers3_weights || !w->skip3_weights)
        return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_create: NULL weight tensor");
    // ---- fold in double: softplus on the convex layers (:238), layer 0 into layer 1
    std::vector<double> A1(NH * 4), W2B((size_t)NH * NH * 4), W2((size_t)NH * NH), S2(NH * 4), w3(NH);
    for (int o = 0; o < NH; ++o) {
        double acc[4] = {w->skip1_weight[o * 3 + 0], w->skip1_weight[o * 3 + 1], w->skip1_weight[o * 3 + 2], w->skip1_bias[o]};
        for (int i = 0; i < NH; ++i) {
            const double wp = softplus_host(w->layers1_weights[o * NH + i]);
            for (int k = 0; k < 3; ++k) acc[k] += wp * w->layers0_weight[i * 3 + k];
            acc[3] += wp * w->layers0_bias[i];
        }
        for (int k = 0; k < 4; ++k) A1[o * 4 + k] = acc[k];
    }
    for (int j = 0; j < NH; ++j) {
        for (int i = 0; i < NH; ++i) {
            const double wp = softplus_host(w->layers2_weights[j * NH + i]);
            W2B[(size_t)j * NH * 4 + i] = wp;
            W2[(size_t)j * NH + i] = wp;
            for (int k = 0; k < 3; ++k) W2B[(size_t)j * NH * 4 + (1 + k) * NH + i] = wp * A1[i * 4 + k];
        }
        for (int k = 0; k < 3; ++k) S2[j * 4 + k] = w->skip2_weight[j * 3 + k];
        S2[j * 4 + 3] = w->skip2_bias[j];
        w3[j] = softplus_host(w->layers3_weights[j]);
    }
    const size_t cnt = A1.size() + W2B.size() + W2.size() + S2.size() + w3.size();
    std::vector<float> h32(cnt);
    std::vector<double> h64(cnt);
    size_t o = 0;
    const size_t oA1 = o; for (double v : A1) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oW = o; for (double v : W2B) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oW2 = o; for (double v : W2) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oS2 = o; for (double v : S2) { h64[o] = v; h32[o++] = (float)v; }
    const size_t ow3 = o; for (double v : w3) { h64[o] = v; h32[o++] = (float)v; }
    dxo_icnn* m = new dxo_icnn();
    hipError_t e = hipSetDevice(ctx->device);
    const size_t bytes64 = cnt * sizeof(double), bytes32 = (cnt * sizeof(float) + 255) / 256 * 256;
    if (e == hipSuccess) e = hipMalloc(&m->dev, bytes64 + bytes32);
    if (e == hipSuccess) e = hipMemcpy(m->dev, h64.data(), bytes64, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy((char*)m->dev + bytes64, h32.data(), cnt * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (m->dev) (void)hipFree(m->dev);
        delete m;
        return dxo_hip_fail(ctx, e, "dxo_icnn_create");
    }
    const double* d64 = (const double*)m->dev;
    const float* d32 = (const float*)((char*)m->dev + bytes64);
    m->f64 = {d64 + oA1, d64 + oW, d64 + oW2, d64 + oS2, d64 + ow3, {0, 0, 0}, {0, 0, 0, 0}};
    m->f32 = {d32 + oA1, d32 + oW, d32 + oW2, d32 + oS2, d32 + ow3, {0, 0, 0}, {0, 0, 0, 0}};
    for (int k = 0; k < 3; ++k) {
        const double s3 = softplus_host(w->skip3_weights[k]);
        m->f64.s3[k] = s3;
        m->f32.s3[k] = (float)s3;
    }
    // ---- H correction (:362-381): H_flat = -P_NN(F = I), with the fp32 network like the reference
    double *dF = nullptr, *ddP = nullptr, *dPp = nullptr;
    const double FI[4] = {1.0, 0.0, 0.0, 1.0};
    double P0[4] = {0, 0, 0, 0};
    e = hipMalloc(&dF, (4 + 16 + 4) * sizeof(double));
    if (e == hipSuccess) {
        ddP = dF + 4;
        dPp = ddP + 16;
        e = hipMemcpy(dF, FI, sizeof FI, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        launch_icnn<float>(m->f32, 1, ctx->stream, (int64_t)1, dF, ddP, dPp);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(P0, dPp, sizeof P0, hipMemcpyDeviceToHost);
    if (dF) (void)hipFree(dF);
    if (e != hipSuccess) {
        (void)hipFree(m->dev);
        delete m;
        return dxo_hip_fail(ctx, e, "dxo_icnn_create: H correction");
    }
    for (int k = 0; k < 4; ++k) {
        m->f32.H[k] = -(double)(float)P0[k];
        m->f64.H[k] = m->f32.H[k];
    }
    *out = m;
    return DXO_OK;
}

extern "C" int dxo_icnn_destroy(dxo_ctx* ctx, dxo_icnn* m) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return DXO_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (m->dev) (void)hipFree(m->dev);
    delete m;
    return DXO_OK;
}

extern "C" int dxo_icnn_correction(dxo_ctx* ctx, const dxo_icnn* m, double* H_flat) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m || !H_flat) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_correction: NULL argument");
    for (int k = 0; k < 4; ++k) H_flat[k] = m->f32.H[k];
    return DXO_OK;
}

extern "C" int dxo_icnn_eval(dxo_ctx* ctx, const dxo_icnn* m, int precision, int64_t n, int mem, const double* F,
                             double* dP, double* P) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_eval: model is NULL");
    if (precision != 0 && precision != 1) return dxo_fail(ctx, DXO_E_OPTION, "dxo_icnn_eval: precision must be 0 (fp32 network) or 1 (fp64)");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_icnn_eval: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_icnn_eval: bad mem");
    if (n > 0 && (!F || !dP || !P)) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_eval: NULL array");
    const uintptr_t all = (uintptr_t)F | (uintptr_t)dP | (uintptr_t)P;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_icnn_eval: arrays must be 8-byte aligned");
    if (mem == DXO_MEM_DEVICE && (all & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_icnn_eval: device arrays must be 16-byte aligned");
    IcnnLaunch L{m, precision};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = icnn_launch(ctx, L, n, F, dP, P, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{F, nullptr, 4 * sd}};
    std::vector<dxo_span> out = {{nullptr, dP, 16 * sd}, {nullptr, P, 4 * sd}};
    return dxo_run_host_pipeline(ctx, n, in, out, icnn_chunk, &L);
}
