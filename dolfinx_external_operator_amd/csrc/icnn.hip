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
// Roofline: compute — the matrix pipe. 192 B/point of HBM traffic against ~45 kflop/point: > 200 flop/B.
// The 64x64 products are a genuine dense contraction (SURVEY.md 8d notes MFMA is legitimate here): five 64x64x64 GEMMs per
// 64-point wave tile. Kernels, by option icnn_variant:
//   2 (DEFAULT)  icnn_mfma_bf16x3: every fp32 operand split exactly into three bf16 numbers, six products per term on
//                v_mfma_f32_32x32x16_bf16 with fp32 accumulate — fp32-level results at 3/8 of the fp32-input pipe time
//                (2.7-2.8 ms per 10^7 points, 0.36 of the bf16 MFMA peak issued)
//   1            icnn_mfma_f32: the same GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32; 4.1 ms, 0.62 of that pipe's peak); cross-check
//   0            icnn_point: lane per point on the vector pipe, weights through the scalar path (s_load into SGPRs); the
//                fp64-network path of BASELINE config 5's tolerance study and the cross-check of both MFMA kernels
#include <cmath>

#include "dxo_common.h"
#include "hyper_core.h"

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
    const T* T1;    // [64][16]: A1_i0 A1_i1 A1_i2 d1_i | 2 A1_i0, 2 A1_i1, 2 A1_i2, 0 | the six products A1_ik A1_il / 6, 0, 0   (icnn_mfma_f32)
    const T* T2;    // [64][8]:  S2_j0 S2_j1 S2_j2 c2_j | w3p_j / 6, 0 0 0                                                         (icnn_mfma_f32)
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

// ------------------------------------------------------------------ fp32 MFMA kernel: helpers
typedef float f32x16 __attribute__((ext_vector_type(16)));

// icnn_mfma_f32 is fully unrolled straight-line code; these keep the compiler from re-ordering it across neurons (it would
// hoist every LDS read and sink every accumulation of a phase, and spill hundreds of registers): a scheduling barrier per
// neuron, and the running sums pinned in registers at the end of each neuron
#ifndef DXO_ICNN_SCHED1
#define DXO_ICNN_SCHED1 __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef DXO_ICNN_SCHED2
#define DXO_ICNN_SCHED2 __builtin_amdgcn_sched_barrier(0);
#endif
#define DXO_ICNN_PIN9(r) asm volatile("" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]));
#define DXO_ICNN_PIN6(r) asm volatile("" : "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]));

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32); }
__device__ __forceinline__ void icnn_lds_fence() {   // wave-private LDS: only the compiler needs ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------ fp32 MFMA kernel, round 3 (variant 1, default)
// Same five GEMMs per 64-point wave tile as icnn_mfma above, restructured so that NOTHING is parked in LDS and no
// softplus is evaluated twice. Per half-tile of 32 points (lane l = (h = l >> 5, p = l & 31), both halves work on
// the same 32 points, each on half of the neurons):
//   phase 1  K-step s = (it, q) of the four forward GEMMs covers the layer-1 neurons i_h(s) = 32 it + (q & 3) + 8 (q >> 2)
//            + 4 h, h = 0, 1 — i.e. lane (h, p) evaluates exactly the neurons whose beta it will later find in ITS OWN
//            accumulator registers (C layout of the beta GEMM): phi''(a1_i) is kept in a register (cph[s]) and the
//            second Hessian term needs neither a second softplus nor a shuffle. A operand sA[jt][s][lane] =
//            W2p[32 jt + p][i_h(s)].
//   phase 2  the layer-2 neurons straight out of the accumulator registers (compile-time indices: everything is
//            unrolled), and every delta_j is fed to the beta GEMM the moment it exists — C layout of the forward GEMMs is
//            the B layout of the beta GEMM — so the 64 beta MFMAs run under the softplus arithmetic of this phase.
//   phase 3  hess += beta_i phi''(a1_i) A1_i A1_i^T from bacc registers x cph registers x a table of the six products.
// LDS per workgroup: A operands 32 KiB + two small tables 6 KiB (rows addressed base(h) + immediate) + 55 KiB of
// lane-private slots for what is only needed after both half-tiles; no accumulator parking, no fences. Round 2's kernel
// spent 36 % of a wave's cycles in vector-only phases (accumulator parking, 128 + 128 LDS accesses per lane and half-tile,
// 32 repeated softplus) and ran at 0.46 of the fp32 MFMA peak; this one issues 6.8 instead of 11.7 vector instructions per
// MFMA and runs at 0.60. What bounds it: the kernel's stream is [vector instructions of a K-step][its MFMAs], a wave issues
// in order, and its two waves per SIMD run in lock step, so the half-tile costs the SUM of its 320 MFMAs (20.5 k cycles) and
// of its vector instructions. (Round 3 read that sum as a hardware property — "vector work does not hide under MFMAs" — from a
// probe whose fillers the compiler had moved behind the MFMAs; the hand-placed probe of round 5, profiles/r05_mfma_gap_probe.txt,
// shows six plain vector instructions riding free behind a 32-cycle MFMA. See DESIGN.md 8.)
__device__ __forceinline__ constexpr int icnn_row(int t, int q) { return 32 * t + (q & 3) + 8 * (q >> 2); }   // + 4 h

// softplus with first and second derivative for icnn_mfma_f32: the exponent is clamped at 80 (e^80 is finite in fp32), so that
// e r = e / (1 + e) and e r r need no select above torch's threshold of 20 — they are 1 and < 2e-9 there, which is what the
// reference's branch returns to fp32 rounding; softplus itself is max(log(1 + e), a): log(1 + e) >= a holds below the
// threshold, and above it the reference returns a.
__device__ __forceinline__ void softplus3_mfma(float a, float& sp, float& s1, float& s2) {
    const float e = __builtin_amdgcn_exp2f(fminf(a, 80.0f) * 1.4426950408889634f);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    const float l = __builtin_amdgcn_logf(1.0f + e) * 0.6931471805599453f;
    sp = fmaxf(l, a);   // below the threshold softplus > a; above it l is a to rounding (or the clamp) and torch returns a
    s1 = e * r;
    s2 = s1 * r;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void icnn_mfma_f32(const float* __restrict__ wT1, const float* __restrict__ wW2,
                                                        const float* __restrict__ wT2, IcnnSmall<float> small, int64_t n,
                                                        const double* __restrict__ F, double* __restrict__ dP,
                                                        double* __restrict__ P) {
    constexpr int BLOCK = WAVES * 64;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = threadIdx.x >> 6;
    __shared__ float sA[2 * 32 * 64];                                      // [jt][s][lane], W2p / 12
    __shared__ float sAT[2 * 32 * 64];                                     // [it][(jt, q)][lane], W2p
    __shared__ __attribute__((aligned(16))) float sT1[NH * 16];            // A1x A1y A1z d1 | 2A1x 2A1y 2A1z 0 | xx xy xz yy | yz zz 0 0  (products / 6)
    __shared__ __attribute__((aligned(16))) float sT2[NH * 8];             // S2x S2y S2z c2 | w3 / 6, 0 0 0
    // per wave, lane-private slots: what is only needed again AFTER the two half-tiles (the fp64 feature derivatives of the
    // lane's own point, 9 doubles) and the finished sums of the half-tile the lane owns (9 floats) wait here instead of
    // occupying 27 registers through the GEMM phases (128 accumulator + 32 beta + 32 phi'' registers are live there)
    __shared__ double sKeep[WAVES * 9 * 64];
    __shared__ float sMine[WAVES * 9 * 64];
    double* keep = sKeep + wave * (9 * 64) + lane;
    float* minep = sMine + wave * (9 * 64) + lane;
    for (int e = threadIdx.x; e < 2 * 32 * 64; e += BLOCK) {
        const int l = e & 63, sq = (e >> 6) & 31, t = e >> 11;
        // the forward GEMMs' A operand carries phi's 1/12: h1 = softplus^2 and u = 2 softplus softplus' enter unscaled
        sA[e] = wW2[(32 * t + (l & 31)) * NH + icnn_row(sq >> 4, sq & 15) + 4 * (l >> 5)] * (1.0f / 12.0f);
        sAT[e] = wW2[(icnn_row(sq >> 4, sq & 15) + 4 * (l >> 5)) * NH + 32 * t + (l & 31)];
    }
    for (int e = threadIdx.x; e < NH * 16; e += BLOCK) sT1[e] = wT1[e];
    for (int e = threadIdx.x; e < NH * 8; e += BLOCK) sT2[e] = wT2[e];
    __syncthreads();
    const float* A_l = sA + lane;
    const float* AT_l = sAT + lane;
    const float* T1_h = sT1 + 4 * h * 16;
    const float* T2_h = sT2 + 4 * h * 8;

    const int64_t n_tiles = (n + 63) / 64;
    const int64_t tile_step = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    // the lane's own point of the NEXT tile is requested one tile ahead: its latency runs under this tile's GEMMs
    dxo_f64x2 f01n{1.0, 0.0}, f23n{0.0, 1.0};
    if (tile < n_tiles) {
        const int64_t pl0 = tile * 64 + lane < n ? tile * 64 + lane : n - 1;
        f01n = reinterpret_cast<const dxo_f64x2*>(F + pl0 * 4)[0];
        f23n = reinterpret_cast<const dxo_f64x2*>(F + pl0 * 4)[1];
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t pidx = tile * 64 + lane;
        float x0, x1, x2;
        {
            const double Fv[4] = {f01n.x, f01n.y, f23n.x, f23n.y};   // tail lanes recompute the last point, never store
            const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
            const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
            const double aD = fabs(D), iD = 1.0 / D;
            const double m = pow(aD, -2.0 / 3.0), nn = m * m;
            x0 = (float)((t + 1.0) * m - 3.0); x1 = (float)((t + D * D) * nn - 3.0); x2 = (float)((aD - 1.0) * (aD - 1.0));
            keep[0 * 64] = Fv[0]; keep[1 * 64] = Fv[1]; keep[2 * 64] = Fv[2]; keep[3 * 64] = Fv[3];
            keep[4 * 64] = t; keep[5 * 64] = D; keep[6 * 64] = m; keep[7 * 64] = nn; keep[8 * 64] = iD;
            asm volatile("" ::: "memory");   // the values are re-read from LDS below, not carried in registers
        }
        if (tile + tile_step < n_tiles) {
            const int64_t pn = (tile + tile_step) * 64 + lane;
            const int64_t pln = pn < n ? pn : n - 1;
            f01n = reinterpret_cast<const dxo_f64x2*>(F + pln * 4)[0];
            f23n = reinterpret_cast<const dxo_f64x2*>(F + pln * 4)[1];
        }
        const float xp0 = xor32(x0), xp1 = xor32(x1), xp2 = xor32(x2);
#pragma unroll 1
        for (int tt = 0; tt < 2; ++tt) {
            const bool own = (tt == h);
            const float xs0 = own ? x0 : xp0, xs1 = own ? x1 : xp1, xs2 = own ? x2 : xp2;
            // the table and A-operand reads below are loop-invariant: without this the compiler hoists hundreds of them out of the
            // half-tile loop into registers (and spills them)
            asm volatile("" ::: "memory");
            float res[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float cph[32];
            f32x16 acc[2][4];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[jt][c][q] = 0.f;
            __builtin_amdgcn_s_setprio(1);   // the MFMA-dense phase goes first on the SIMD's issue port (-3 % measured)
            // ---- phase 1: [a2 | g0 | g1 | g2] = (W2p / 12) @ [sp^2 | u 2A1_0 | u 2A1_1 | u 2A1_2],  u = sp sp'
            // One K-step's B operands (this lane's layer-1 neuron) and A operands:
            struct Ops { float Ar0, Ar1, hv, B1, B2, B3; };
            auto step_ops = [&](int s, float& cph_s) -> Ops {
                const float4 a1 = *reinterpret_cast<const float4*>(T1_h + icnn_row(s >> 4, s & 15) * 16);
                const float4 b1 = *reinterpret_cast<const float4*>(T1_h + icnn_row(s >> 4, s & 15) * 16 + 4);
                Ops o;
                o.Ar0 = A_l[(0 * 32 + s) * 64];
                o.Ar1 = A_l[(1 * 32 + s) * 64];
                const float a = fmaf(a1.x, xs0, fmaf(a1.y, xs1, fmaf(a1.z, xs2, a1.w)));
                float sp, s1, s2;
                softplus3_mfma(a, sp, s1, s2);
                o.hv = sp * sp;
                const float uv = sp * s1;
                cph_s = fmaf(sp, s2, s1 * s1);     // phi''(a1) * 6: the 1/6 sits in the table of products (phase 3)
                asm volatile("" : "+v"(cph_s));    // materialise it HERE: otherwise its computation is sunk into phase 3 and its four inputs spilled
                o.B1 = uv * b1.x; o.B2 = uv * b1.y; o.B3 = uv * b1.z;
                return o;
            };
            auto step_mfma = [&](const Ops& o) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar0, o.hv, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar1, o.hv, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar0, o.B1, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar1, o.B1, acc[1][1], 0, 0, 0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar0, o.B2, acc[0][2], 0, 0, 0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar1, o.B2, acc[1][2], 0, 0, 0);
                acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar0, o.B3, acc[0][3], 0, 0, 0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.Ar1, o.B3, acc[1][3], 0, 0, 0);
            };
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const Ops o = step_ops(s, cph[s]);
                step_mfma(o);
                DXO_ICNN_SCHED1
            }
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 2: layer-2 neurons from the accumulators; delta feeds beta = W2p^T delta at once
            f32x16 bacc[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) bacc[it][q] = 0.f;
#pragma unroll
            for (int jq = 0; jq < 32; ++jq) {
                const int jt = jq >> 4, q = jq & 15;
                const float4 s2r = *reinterpret_cast<const float4*>(T2_h + icnn_row(jt, q) * 8);
                const float w6 = T2_h[icnn_row(jt, q) * 8 + 4];
                const float At0 = AT_l[(0 * 32 + jq) * 64], At1 = AT_l[(1 * 32 + jq) * 64];
                const float a2 = acc[jt][0][q] + fmaf(s2r.x, xs0, fmaf(s2r.y, xs1, fmaf(s2r.z, xs2, s2r.w)));
                const float g0 = acc[jt][1][q] + s2r.x, g1 = acc[jt][2][q] + s2r.y, g2 = acc[jt][3][q] + s2r.z;
                float sp, s1, s2;
                softplus3_mfma(a2, sp, s1, s2);
                const float delta = w6 * sp * s1;
                const float curv = w6 * fmaf(sp, s2, s1 * s1);
                bacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(At0, delta, bacc[0], 0, 0, 0);
                bacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(At1, delta, bacc[1], 0, 0, 0);
                const float cg0 = curv * g0, cg1 = curv * g1, cg2 = curv * g2;
                res[0] = fmaf(delta, g0, res[0]); res[1] = fmaf(delta, g1, res[1]); res[2] = fmaf(delta, g2, res[2]);
                res[3] = fmaf(cg0, g0, res[3]); res[4] = fmaf(cg0, g1, res[4]); res[5] = fmaf(cg0, g2, res[5]);
                res[6] = fmaf(cg1, g1, res[6]); res[7] = fmaf(cg1, g2, res[7]); res[8] = fmaf(cg2, g2, res[8]);
                // the nine sums are finished HERE (otherwise their updates are sunk below the whole unrolled phase and g, curv,
                // delta of every neuron stay live: hundreds of spilled registers)
                DXO_ICNN_PIN9(res)
                DXO_ICNN_SCHED2
            }
            // ---- phase 3: second Hessian term, beta_i (accumulator registers) x phi''(a1_i) (cph) x A1_i A1_i^T / 6 (table)
#pragma unroll
            for (int iq = 0; iq < 32; ++iq) {
                const int it = iq >> 4, q = iq & 15;
                const float4 pa = *reinterpret_cast<const float4*>(T1_h + icnn_row(it, q) * 16 + 8);
                const float2 pb = *reinterpret_cast<const float2*>(T1_h + icnn_row(it, q) * 16 + 12);
                const float c = bacc[it][q] * cph[iq];
                res[3] = fmaf(c, pa.x, res[3]); res[4] = fmaf(c, pa.y, res[4]); res[5] = fmaf(c, pa.z, res[5]);
                res[6] = fmaf(c, pa.w, res[6]); res[7] = fmaf(c, pb.x, res[7]); res[8] = fmaf(c, pb.y, res[8]);
                DXO_ICNN_PIN6(res)
                DXO_ICNN_SCHED2
            }
            // both half-waves hold partial sums of the SAME 32 points: add them, keep the tile this lane owns
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const float tot = res[q] + xor32(res[q]);
                if (own) minep[q * 64] = tot;
            }
        }
        asm volatile("" ::: "memory");
        if (pidx < n) {
            const double Fk[4] = {keep[0 * 64], keep[1 * 64], keep[2 * 64], keep[3 * 64]};
            const double t = keep[4 * 64], D = keep[5 * 64], m = keep[6 * 64], nn = keep[7 * 64], iD = keep[8 * 64];
            const double aD = fabs(D), sg = D < 0.0 ? -1.0 : 1.0;
            float mine[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) mine[q] = minep[q * 64];
            const double kt[3] = {m, nn, 0.0};
            const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                                  2.0 * (aD - 1.0) * sg};
            const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
            const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                                   -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
            const float y1f[3] = {mine[0] + small.s3[0], mine[1] + small.s3[1], mine[2] + small.s3[2]};
            icnn_chain(Fk, kt, kD, ktD, kDD, y1f, mine + 3, small.H, dP + pidx * 16, P + pidx * 4);
        }
    }
}

// ------------------------------------------------------------------ fp32-equivalent MFMA kernel on the bf16 pipe (icnn_variant = 2, the default)
// The same five GEMMs with every fp32 operand split EXACTLY into three bf16 numbers (8 + 8 + 8 mantissa bits by
// truncation: v = h + m + l) and the product formed as h h' + h m' + m h' + h l' + l h' + m m' on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the dropped terms are below 2^-24 of the product, i.e. at fp32 rounding
// level, while six bf16 MFMAs (16 x the fp32-input rate each) cost 3/8 of the matrix-pipe time of the fp32 form.
// K-step s of the forward GEMMs covers 16 layer-1 neurons: lane (h, p) evaluates the eight neurons
// icnn_row(s >> 1, 8 (s & 1) + i) + 4 h, i = 0..7, of ITS point and packs their split parts as the B fragment (k = 8 h + i);
// the enumeration over s is again the C layout of the beta GEMM, so beta_i meets phi''(a1_i) in the same lane. A fragments
// (the split weights, eight bf16 per lane) are prepared once per workgroup in LDS.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 icnn_bf16x8;
typedef unsigned int icnn_u32x4 __attribute__((ext_vector_type(4)));

struct IcnnSplit { icnn_u32x4 h, m, l; };
struct IcnnSplitW { unsigned h[4], m[4], l[4]; };   // the same, dword-addressable while it is being filled

__device__ __forceinline__ void icnn_split1(float v, unsigned& hb, unsigned& mb, unsigned& lb) {
    hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    mb = __float_as_uint(r1) & 0xffff0000u;
    lb = __float_as_uint(r1 - __uint_as_float(mb));   // at most 8 significant bits left: a bf16 number, low half zero
}

__device__ __forceinline__ f32x16 icnn_mfma_bf16(icnn_u32x4 a, icnn_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(icnn_bf16x8, a), __builtin_bit_cast(icnn_bf16x8, b), c, 0, 0, 0);
}

// A neuron PAIR's two fp32 values. A plain struct with scalar operators, NOT an ext_vector_type: <2 x float> arithmetic becomes
// v_pk_{fma,mul,add}_f32, and a packed fp32 instruction issued beside a running MFMA stalls ~20 cycles
// (profiles/r05_mfma_gap_probe.txt: one v_pk_fma_f32 per 32-cycle MFMA gap: 52 cycles per MFMA; six v_fma_f32: 37;
// MI355X_MICROARCH.md: "packed f32 VALU ... an anti-lever beside MFMAs"). The arithmetic is identical (IEEE fp32 either way);
// this translation unit is built with -fno-slp-vectorize so that the compiler does not re-pack adjacent scalar operations
// (_build.py, Makefile).
struct icnn_f2 {
    float x, y;
};
__device__ __forceinline__ icnn_f2 operator+(icnn_f2 a, icnn_f2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ icnn_f2 operator-(icnn_f2 a, icnn_f2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ icnn_f2 operator*(icnn_f2 a, icnn_f2 b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ icnn_f2 operator+(icnn_f2 a, float b) { return {a.x + b, a.y + b}; }
__device__ __forceinline__ icnn_f2 operator*(icnn_f2 a, float b) { return {a.x * b, a.y * b}; }

__device__ __forceinline__ IcnnSplit icnn_pack(const IcnnSplitW& w) {
    IcnnSplit o;
    o.h = icnn_u32x4{w.h[0], w.h[1], w.h[2], w.h[3]};
    o.m = icnn_u32x4{w.m[0], w.m[1], w.m[2], w.m[3]};
    o.l = icnn_u32x4{w.l[0], w.l[1], w.l[2], w.l[3]};
    return o;
}

// one B-fragment dword (two consecutive k: the neuron pair) of each part from a pair of fp32 values
__device__ __forceinline__ void icnn_split_pair(icnn_f2 v, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned vx = __float_as_uint(v.x), vy = __float_as_uint(v.y);
    h = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    const icnn_f2 vh = {__uint_as_float(vx & 0xffff0000u), __uint_as_float(vy & 0xffff0000u)};
    const icnn_f2 r = v - vh;
    const unsigned rx = __float_as_uint(r.x), ry = __float_as_uint(r.y);
    m = __builtin_amdgcn_perm(ry, rx, 0x07060302u);
    const icnn_f2 rh = {__uint_as_float(rx & 0xffff0000u), __uint_as_float(ry & 0xffff0000u)};
    const icnn_f2 q = r - rh;
    l = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x07060302u);
}

// softplus, softplus', softplus'' of two pre-activations: softplus3_mfma's operations, the multiplies and adds as packed fp32
__device__ __forceinline__ void softplus3_pk(icnn_f2 a, icnn_f2& sp, icnn_f2& s1, icnn_f2& s2) {
    const icnn_f2 am = {fminf(a.x, 80.0f), fminf(a.y, 80.0f)};
    const icnn_f2 t = am * 1.4426950408889634f;
    const icnn_f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const icnn_f2 ope = e + 1.0f;
    const icnn_f2 r = {__builtin_amdgcn_rcpf(ope.x), __builtin_amdgcn_rcpf(ope.y)};
    const icnn_f2 l = icnn_f2{__builtin_amdgcn_logf(ope.x), __builtin_amdgcn_logf(ope.y)} * 0.6931471805599453f;
    sp = icnn_f2{fmaxf(l.x, a.x), fmaxf(l.y, a.y)};
    s1 = e * r;
    s2 = s1 * r;
}

__device__ __forceinline__ icnn_f2 icnn_fma2(icnn_f2 a, icnn_f2 b, icnn_f2 c) { return {__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }

// c0, c1 += A[jt] (split) x B (split): the six products with a weight above 2^-24, smallest first, accumulators alternating
__device__ __forceinline__ void icnn_mma6(const IcnnSplit (&A)[2], const IcnnSplit& B, f32x16& c0, f32x16& c1) {
    c0 = icnn_mfma_bf16(A[0].l, B.h, c0); c1 = icnn_mfma_bf16(A[1].l, B.h, c1);
    c0 = icnn_mfma_bf16(A[0].h, B.l, c0); c1 = icnn_mfma_bf16(A[1].h, B.l, c1);
    c0 = icnn_mfma_bf16(A[0].m, B.m, c0); c1 = icnn_mfma_bf16(A[1].m, B.m, c1);
    c0 = icnn_mfma_bf16(A[0].m, B.h, c0); c1 = icnn_mfma_bf16(A[1].m, B.h, c1);
    c0 = icnn_mfma_bf16(A[0].h, B.m, c0); c1 = icnn_mfma_bf16(A[1].h, B.m, c1);
    c0 = icnn_mfma_bf16(A[0].h, B.h, c0); c1 = icnn_mfma_bf16(A[1].h, B.h, c1);
}

// not `volatile`: a volatile asm orders every LDS read around it and would pin the table reads between the pins
#define DXO_ICNN_PIN2(PAIR_) asm("" : "+v"((PAIR_).x), "+v"((PAIR_).y));
#ifndef DXO_ICNN_PIPE_AGPR
#define DXO_ICNN_PIPE_AGPR false
#endif

// ------------------------------------------------------------------ split-bf16 kernel, phases in sequence, PACKED fp32 vector arithmetic (icnn_variant = 2, the default)
// Rounds 3-4's kernel, unchanged: two waves per SIMD, every neuron pair's arithmetic as v_pk_{fma,mul,add}_f32. A packed fp32
// instruction cannot run beside an MFMA (profiles/r05_mfma_gap_probe.txt: it waits for the MFMA in flight and ~16 cycles more),
// so this kernel's half-tile costs the SUM of its matrix-pipe and vector cycles — but its vector work is 1 770 instructions
// where the scalar form below needs 2 850, which is why it is still the faster of the two (profiles/NOTES_icnn.md).
typedef float icnn_f2p __attribute__((ext_vector_type(2)));

// one B-fragment dword (two consecutive k: the neuron pair) of each part from a pair of fp32 values
__device__ __forceinline__ void icnn_split_pair_p(icnn_f2p v, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned vx = __float_as_uint(v.x), vy = __float_as_uint(v.y);
    h = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    const icnn_f2p vh = {__uint_as_float(vx & 0xffff0000u), __uint_as_float(vy & 0xffff0000u)};
    const icnn_f2p r = v - vh;
    const unsigned rx = __float_as_uint(r.x), ry = __float_as_uint(r.y);
    m = __builtin_amdgcn_perm(ry, rx, 0x07060302u);
    const icnn_f2p rh = {__uint_as_float(rx & 0xffff0000u), __uint_as_float(ry & 0xffff0000u)};
    const icnn_f2p q = r - rh;
    l = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x07060302u);
}


// softplus, softplus', softplus'' of two pre-activations: softplus3_mfma's operations, the multiplies and adds as packed fp32
__device__ __forceinline__ void softplus3_p(icnn_f2p a, icnn_f2p& sp, icnn_f2p& s1, icnn_f2p& s2) {
    const icnn_f2p am = {fminf(a.x, 80.0f), fminf(a.y, 80.0f)};
    const icnn_f2p t = am * 1.4426950408889634f;
    const icnn_f2p e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const icnn_f2p ope = e + 1.0f;
    const icnn_f2p r = {__builtin_amdgcn_rcpf(ope.x), __builtin_amdgcn_rcpf(ope.y)};
    const icnn_f2p l = icnn_f2p{__builtin_amdgcn_logf(ope.x), __builtin_amdgcn_logf(ope.y)} * 0.6931471805599453f;
    sp = icnn_f2p{fmaxf(l.x, a.x), fmaxf(l.y, a.y)};
    s1 = e * r;
    s2 = s1 * r;
}


__device__ __forceinline__ icnn_f2p icnn_fma2_p(icnn_f2p a, icnn_f2p b, icnn_f2p c) { return __builtin_elementwise_fma(a, b, c); }

#define DXO_ICNN_PINP(x) asm volatile("" : "+v"(x));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void icnn_mfma_bf16x3_packed(const float* __restrict__ wT1, const float* __restrict__ wW2,
                                                        const float* __restrict__ wT2, IcnnSmall<float> small, int64_t n,
                                                        const double* __restrict__ F, double* __restrict__ dP,
                                                        double* __restrict__ P) {
    constexpr int BLOCK = WAVES * 64;
    constexpr int FRAG = 64 * 8;   // bf16 per fragment (64 lanes x 8)
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = threadIdx.x >> 6;
    // A fragments, [matrix][t][step][part][lane] x 8 bf16 (one 16-byte read per lane and fragment). Forward matrices:
    //   c = 0: W2p / 12 (meets h1 = softplus^2);  c = 1..3: W2p[j][i] / 12 * 2 A1[i][c - 1] (meets u = softplus softplus'):
    // the layer-1 input weights ride in the A operand, so all three gradient GEMMs share ONE B operand.
    __shared__ __attribute__((aligned(16))) unsigned short sA3[4 * 2 * 4 * 3 * FRAG];
    __shared__ __attribute__((aligned(16))) unsigned short sAT3[2 * 4 * 3 * FRAG];   // beta GEMM: W2p transposed
    // per neuron PAIR (rows 2 r, 2 r + 1; the two values of a quantity adjacent: operands of the packed fp32 instructions)
    __shared__ __attribute__((aligned(16))) float sP1[32 * 8];    // A1x A1y A1z d1
    __shared__ __attribute__((aligned(16))) float sP3[32 * 12];   // the six products A1_k A1_l / 6
    __shared__ __attribute__((aligned(16))) float sP2[32 * 12];   // S2x S2y S2z c2 w3/6, 0
    __shared__ float sMine[WAVES * 9 * 64];
    float* minep = sMine + wave * (9 * 64) + lane;
    for (int e = threadIdx.x; e < 2 * 4 * 64 * 8; e += BLOCK) {
        const int i = e & 7, l = (e >> 3) & 63, st = (e >> 9) & 3, t = e >> 11;
        const int row = icnn_row(st >> 1, 8 * (st & 1) + i) + 4 * (l >> 5);
        const float wf = wW2[(32 * t + (l & 31)) * NH + row] * (1.0f / 12.0f);   // forward GEMMs: A[j = 32 t + p][k -> neuron row]
        const float wb = wW2[row * NH + 32 * t + (l & 31)];                      // beta GEMM:     A[i = 32 t + p][k -> neuron row (a j)]
        const int base = ((t * 4 + st) * 3) * FRAG + l * 8 + i;
        unsigned hb, mb, lb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            icnn_split1(c == 0 ? wf : wf * wT1[row * 16 + 3 + c], hb, mb, lb);
            unsigned short* d = sA3 + c * (2 * 4 * 3 * FRAG) + base;
            d[0] = (unsigned short)(hb >> 16); d[FRAG] = (unsigned short)(mb >> 16); d[2 * FRAG] = (unsigned short)(lb >> 16);
        }
        icnn_split1(wb, hb, mb, lb);
        sAT3[base] = (unsigned short)(hb >> 16); sAT3[base + FRAG] = (unsigned short)(mb >> 16); sAT3[base + 2 * FRAG] = (unsigned short)(lb >> 16);
    }
    for (int e = threadIdx.x; e < NH * 4; e += BLOCK) {
        const int row = e >> 2, k = e & 3;
        sP1[(row >> 1) * 8 + 2 * k + (row & 1)] = wT1[row * 16 + k];
    }
    for (int e = threadIdx.x; e < NH * 6; e += BLOCK) {
        const int row = e / 6, k = e % 6;
        sP3[(row >> 1) * 12 + 2 * k + (row & 1)] = wT1[row * 16 + 8 + k];
        sP2[(row >> 1) * 12 + 2 * k + (row & 1)] = k < 5 ? wT2[row * 8 + k] : 0.0f;
    }
    __syncthreads();
    const icnn_u32x4* A_l = reinterpret_cast<const icnn_u32x4*>(sA3) + lane;     // fragment (c, t, st, part) at (((c*2+t)*4+st)*3+part)*64
    const icnn_u32x4* AT_l = reinterpret_cast<const icnn_u32x4*>(sAT3) + lane;
    // the lane's neurons are rows icnn_row(t, q) + 4 h: pair index (icnn_row(t, q) >> 1) + 2 h for even q
    const float* P1_h = sP1 + 2 * h * 8;
    const float* P3_h = sP3 + 2 * h * 12;
    const float* P2_h = sP2 + 2 * h * 12;

    const int64_t n_tiles = (n + 63) / 64;
    const int64_t tile_step = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    dxo_f64x2 f01n{1.0, 0.0}, f23n{0.0, 1.0};
    if (tile < n_tiles) {
        const int64_t pl0 = tile * 64 + lane < n ? tile * 64 + lane : n - 1;
        f01n = reinterpret_cast<const dxo_f64x2*>(F + pl0 * 4)[0];
        f23n = reinterpret_cast<const dxo_f64x2*>(F + pl0 * 4)[1];
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t pidx = tile * 64 + lane;
        const dxo_f64x2 f01 = f01n, f23 = f23n;
        float x0, x1, x2;
        {
            const double Fv[4] = {f01.x, f01.y, f23.x, f23.y};   // tail lanes recompute the last point, never store
            const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
            const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
            const double aD = fabs(D);
            const double m = hyper_pow_m23(aD), nn = m * m;
            x0 = (float)((t + 1.0) * m - 3.0); x1 = (float)((t + D * D) * nn - 3.0); x2 = (float)((aD - 1.0) * (aD - 1.0));
        }
        if (tile + tile_step < n_tiles) {
            const int64_t pn = (tile + tile_step) * 64 + lane;
            const int64_t pln = pn < n ? pn : n - 1;
            f01n = reinterpret_cast<const dxo_f64x2*>(F + pln * 4)[0];
            f23n = reinterpret_cast<const dxo_f64x2*>(F + pln * 4)[1];
        }
        const float xp0 = xor32(x0), xp1 = xor32(x1), xp2 = xor32(x2);
#pragma unroll 1
        for (int tt = 0; tt < 2; ++tt) {
            const bool own = (tt == h);
            const float xs0 = own ? x0 : xp0, xs1 = own ? x1 : xp1, xs2 = own ? x2 : xp2;
            asm volatile("" ::: "memory");   // keep the loop-invariant LDS reads inside the loop (see icnn_mfma_f32)
            icnn_f2p res[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) res[q] = icnn_f2p{0.f, 0.f};
            icnn_f2p cph[16];
            f32x16 acc[2][4];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[jt][c][q] = 0.f;
            // ---- phase 1: four K-steps of 16 layer-1 neurons (eight per lane, as four pairs)
            //      [a2 | g0 | g1 | g2] += [A_0 h1 | A_1 u | A_2 u | A_3 u]
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                IcnnSplitW Bhw, Buw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pr = icnn_row(st >> 1, 8 * (st & 1) + 2 * k) >> 1;
                    const float4 ta = *reinterpret_cast<const float4*>(P1_h + pr * 8);       // A1x pair, A1y pair
                    const float4 tb = *reinterpret_cast<const float4*>(P1_h + pr * 8 + 4);   // A1z pair, d1 pair
                    const icnn_f2p a = icnn_fma2_p(icnn_f2p{ta.x, ta.y}, icnn_f2p{xs0, xs0},
                                                icnn_fma2_p(icnn_f2p{ta.z, ta.w}, icnn_f2p{xs1, xs1},
                                                          icnn_fma2_p(icnn_f2p{tb.x, tb.y}, icnn_f2p{xs2, xs2}, icnn_f2p{tb.z, tb.w})));
                    icnn_f2p sp, s1, s2;
                    softplus3_p(a, sp, s1, s2);
                    const icnn_f2p hv = sp * sp, uv = sp * s1;
                    cph[st * 4 + k] = icnn_fma2_p(sp, s2, s1 * s1);   // phi''(a1) * 6: the 1/6 sits in the table of products (phase 3)
                    DXO_ICNN_PINP(cph[st * 4 + k])
                    icnn_split_pair_p(hv, Bhw.h[k], Bhw.m[k], Bhw.l[k]);
                    icnn_split_pair_p(uv, Buw.h[k], Buw.m[k], Buw.l[k]);
                }
                const IcnnSplit Bh = icnn_pack(Bhw), Bu = icnn_pack(Buw);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    IcnnSplit A[2];
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) {
                        A[jt].h = A_l[(((c * 2 + jt) * 4 + st) * 3 + 0) * 64];
                        A[jt].m = A_l[(((c * 2 + jt) * 4 + st) * 3 + 1) * 64];
                        A[jt].l = A_l[(((c * 2 + jt) * 4 + st) * 3 + 2) * 64];
                    }
                    icnn_mma6(A, c == 0 ? Bh : Bu, acc[0][c], acc[1][c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase 2: layer-2 neurons in groups of eight (four pairs); each group's deltas are one B fragment of the beta GEMM
            f32x16 bacc[2];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int q = 0; q < 16; ++q) bacc[it][q] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int jt = g >> 1;
                IcnnSplitW Bdw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = 8 * (g & 1) + 2 * k;
                    const int pr = icnn_row(jt, q) >> 1;
                    const float4 ta = *reinterpret_cast<const float4*>(P2_h + pr * 12);       // S2x pair, S2y pair
                    const float4 tb = *reinterpret_cast<const float4*>(P2_h + pr * 12 + 4);   // S2z pair, c2 pair
                    const float2 tw = *reinterpret_cast<const float2*>(P2_h + pr * 12 + 8);   // w3 / 6 pair
                    const icnn_f2p S2x = {ta.x, ta.y}, S2y = {ta.z, ta.w}, S2z = {tb.x, tb.y}, w6 = {tw.x, tw.y};
                    const icnn_f2p a2 = icnn_f2p{acc[jt][0][q], acc[jt][0][q + 1]} +
                                       icnn_fma2_p(S2x, icnn_f2p{xs0, xs0}, icnn_fma2_p(S2y, icnn_f2p{xs1, xs1}, icnn_fma2_p(S2z, icnn_f2p{xs2, xs2}, icnn_f2p{tb.z, tb.w})));
                    const icnn_f2p g0 = icnn_f2p{acc[jt][1][q], acc[jt][1][q + 1]} + S2x;
                    const icnn_f2p g1 = icnn_f2p{acc[jt][2][q], acc[jt][2][q + 1]} + S2y;
                    const icnn_f2p g2 = icnn_f2p{acc[jt][3][q], acc[jt][3][q + 1]} + S2z;
                    icnn_f2p sp, s1, s2;
                    softplus3_p(a2, sp, s1, s2);
                    const icnn_f2p dl = w6 * sp * s1;
                    const icnn_f2p curv = w6 * icnn_fma2_p(sp, s2, s1 * s1);
                    icnn_split_pair_p(dl, Bdw.h[k], Bdw.m[k], Bdw.l[k]);
                    const icnn_f2p cg0 = curv * g0, cg1 = curv * g1, cg2 = curv * g2;
                    res[0] = icnn_fma2_p(dl, g0, res[0]); res[1] = icnn_fma2_p(dl, g1, res[1]); res[2] = icnn_fma2_p(dl, g2, res[2]);
                    res[3] = icnn_fma2_p(cg0, g0, res[3]); res[4] = icnn_fma2_p(cg0, g1, res[4]); res[5] = icnn_fma2_p(cg0, g2, res[5]);
                    res[6] = icnn_fma2_p(cg1, g1, res[6]); res[7] = icnn_fma2_p(cg1, g2, res[7]); res[8] = icnn_fma2_p(cg2, g2, res[8]);
#pragma unroll
                    for (int r = 0; r < 9; ++r) DXO_ICNN_PINP(res[r])
                }
                IcnnSplit A[2];
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    A[it].h = AT_l[((it * 4 + g) * 3 + 0) * 64];
                    A[it].m = AT_l[((it * 4 + g) * 3 + 1) * 64];
                    A[it].l = AT_l[((it * 4 + g) * 3 + 2) * 64];
                }
                icnn_mma6(A, icnn_pack(Bdw), bacc[0], bacc[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- phase 3: second Hessian term, beta_i (accumulator registers) x phi''(a1_i) (cph) x A1_i A1_i^T / 6 (table)
#pragma unroll
            for (int ip = 0; ip < 16; ++ip) {
                const int it = ip >> 3, q = 2 * (ip & 7);
                const int pr = icnn_row(it, q) >> 1;
                const float4 pa = *reinterpret_cast<const float4*>(P3_h + pr * 12);
                const float4 pb = *reinterpret_cast<const float4*>(P3_h + pr * 12 + 4);
                const float4 pc = *reinterpret_cast<const float4*>(P3_h + pr * 12 + 8);
                const icnn_f2p c = icnn_f2p{bacc[it][q], bacc[it][q + 1]} * cph[ip];
                res[3] = icnn_fma2_p(c, icnn_f2p{pa.x, pa.y}, res[3]); res[4] = icnn_fma2_p(c, icnn_f2p{pa.z, pa.w}, res[4]);
                res[5] = icnn_fma2_p(c, icnn_f2p{pb.x, pb.y}, res[5]); res[6] = icnn_fma2_p(c, icnn_f2p{pb.z, pb.w}, res[6]);
                res[7] = icnn_fma2_p(c, icnn_f2p{pc.x, pc.y}, res[7]); res[8] = icnn_fma2_p(c, icnn_f2p{pc.z, pc.w}, res[8]);
#pragma unroll
                for (int r = 3; r < 9; ++r) DXO_ICNN_PINP(res[r])
            }
            // both half-waves hold partial sums of the SAME 32 points: add them, keep the tile this lane owns
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const float part = res[q].x + res[q].y;
                const float tot = part + xor32(part);
                if (own) minep[q * 64] = tot;
            }
        }
        asm volatile("" ::: "memory");
        if (pidx < n) {
            // the fp64 feature derivatives are formed here, after the GEMM phases (they would occupy 18 registers through them)
            const double Fk[4] = {f01.x, f01.y, f23.x, f23.y};
            const double t = Fk[0] * Fk[0] + Fk[1] * Fk[1] + Fk[2] * Fk[2] + Fk[3] * Fk[3];
            const double D = Fk[0] * Fk[3] - Fk[1] * Fk[2];
            const double aD = fabs(D), sg = D < 0.0 ? -1.0 : 1.0, iD = 1.0 / D;
            const double m = hyper_pow_m23(aD), nn = m * m;
            float mine[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) mine[q] = minep[q * 64];
            const double kt[3] = {m, nn, 0.0};
            const double kD[3] = {(t + 1.0) * (-2.0 / 3.0) * m * iD, 2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn * iD,
                                  2.0 * (aD - 1.0) * sg};
            const double ktD[3] = {(-2.0 / 3.0) * m * iD, (-4.0 / 3.0) * nn * iD, 0.0};
            const double kDD[3] = {(t + 1.0) * (10.0 / 9.0) * m * iD * iD,
                                   -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn * iD * iD, 2.0};
            const float y1f[3] = {mine[0] + small.s3[0], mine[1] + small.s3[1], mine[2] + small.s3[2]};
            icnn_chain(Fk, kt, kD, ktD, kDD, y1f, mine + 3, small.H, dP + pidx * 16, P + pidx * 4);
        }
    }
}


#ifdef DXO_EXPERIMENTS      // icnn_mfma_bf16x3_pipe / _hybrid (icnn_variant 3 / 4): measured, bit-identical, slower — not shipped
#include "../../scripts/exp/icnn_variants.h"
#endif

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
        // MFMA kernel: one wave = 64 points, persistent grid of one 8-wave workgroup per CU (two waves per SIMD: the register
        // budget of 256 per lane is what the 128 + 32 accumulator registers need)
        IcnnSmall<float> small;
        for (int k = 0; k < 3; ++k) small.s3[k] = L.m->f32.s3[k];
        for (int k = 0; k < 4; ++k) small.H[k] = L.m->f32.H[k];
        int64_t mb = (n + 8 * 64 - 1) / (8 * 64);
        if (mb > ctx->compute_units) mb = ctx->compute_units;
#ifdef DXO_EXPERIMENTS
        if (ctx->icnn_variant == 4 || ctx->icnn_variant == 3) {   // scripts/exp/icnn_variants.h: one 4-wave workgroup per CU (one wave per SIMD)
            int64_t mp = (n + 4 * 64 - 1) / (4 * 64);
            if (mp > ctx->compute_units) mp = ctx->compute_units;
            if (ctx->icnn_variant == 4)
                hipLaunchKernelGGL((icnn_mfma_bf16x3_hybrid<4>), dim3((int)mp), dim3(256), 0, s, L.m->f32.T1, L.m->f32.W2, L.m->f32.T2, small, n, F, dP, P);
            else
                hipLaunchKernelGGL((icnn_mfma_bf16x3_pipe<4>), dim3((int)mp), dim3(256), 0, s, L.m->f32.T1, L.m->f32.W2, L.m->f32.T2, small, n, F, dP, P);
        } else
#endif
        if (ctx->icnn_variant != 1)   // 2: split-bf16 products phase after phase, two waves per SIMD; 1: fp32-input MFMA
            hipLaunchKernelGGL((icnn_mfma_bf16x3_packed<8>), dim3((int)mb), dim3(512), 0, s, L.m->f32.T1, L.m->f32.W2, L.m->f32.T2, small, n, F, dP, P);
        else
            hipLaunchKernelGGL((icnn_mfma_f32<8>), dim3((int)mb), dim3(512), 0, s, L.m->f32.T1, L.m->f32.W2, L.m->f32.T2, small, n, F, dP, P);
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
        isihara_point(prm, Fv, Xd + lane * 16, Xp + lane * 4);
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

int dxo_icnn_variant_max() {
#ifdef DXO_EXPERIMENTS
    return 4;
#else
    return 2;
#endif
}

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
    return dxo_run_host_pipeline(ctx, n, in, out, isihara_chunk, &k, 1, nullptr, true);
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
    // tables of the round-3 MFMA kernel (products of the fp32-rounded A1, as the kernel would form them)
    std::vector<double> T1(NH * 16, 0.0), T2(NH * 8, 0.0);
    for (int i = 0; i < NH; ++i) {
        const double ax = (double)(float)A1[i * 4 + 0], ay = (double)(float)A1[i * 4 + 1], az = (double)(float)A1[i * 4 + 2];
        const double row[14] = {A1[i * 4 + 0], A1[i * 4 + 1], A1[i * 4 + 2], A1[i * 4 + 3], 2 * ax, 2 * ay, 2 * az, 0.0,
                                ax * ax / 6, ax * ay / 6, ax * az / 6, ay * ay / 6, ay * az / 6, az * az / 6};
        for (int k = 0; k < 14; ++k) T1[i * 16 + k] = row[k];
        for (int k = 0; k < 4; ++k) T2[i * 8 + k] = S2[i * 4 + k];
        T2[i * 8 + 4] = w3[i] / 6;
    }
    const size_t cnt = A1.size() + W2B.size() + W2.size() + S2.size() + w3.size() + T1.size() + T2.size();
    std::vector<float> h32(cnt);
    std::vector<double> h64(cnt);
    size_t o = 0;
    const size_t oA1 = o; for (double v : A1) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oW = o; for (double v : W2B) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oW2 = o; for (double v : W2) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oS2 = o; for (double v : S2) { h64[o] = v; h32[o++] = (float)v; }
    const size_t ow3 = o; for (double v : w3) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oT1 = o; for (double v : T1) { h64[o] = v; h32[o++] = (float)v; }
    const size_t oT2 = o; for (double v : T2) { h64[o] = v; h32[o++] = (float)v; }
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
    m->f64 = {d64 + oA1, d64 + oW, d64 + oW2, d64 + oS2, d64 + ow3, d64 + oT1, d64 + oT2, {0, 0, 0}, {0, 0, 0, 0}};
    m->f32 = {d32 + oA1, d32 + oW, d32 + oW2, d32 + oS2, d32 + ow3, d32 + oT1, d32 + oT2, {0, 0, 0}, {0, 0, 0, 0}};
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

// internal: the kernels on device pointers and an explicit stream (field_ops.hip)
int dxo_icnn_launch_device(dxo_ctx* ctx, const dxo_icnn* m, int precision, int64_t n, const double* F, double* dP, double* P,
                           hipStream_t s) {
    IcnnLaunch L{m, precision};
    return icnn_launch(ctx, L, n, F, dP, P, s);
}

int dxo_isihara_launch_device(dxo_ctx* ctx, const dxo_isihara_params* prm, int64_t n, const double* F, double* dP, double* P,
                              hipStream_t s) {
    return isihara_launch(ctx, IsiPrm{prm->c1, prm->c2, prm->c3, prm->c4}, n, F, dP, P, s);
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
