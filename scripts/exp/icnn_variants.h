// icnn_variants.h — ICNN kernels that were measured and NOT shipped (round 5; DESIGN.md 8, profiles/r05_icnn_pipe_pmc.txt,
// profiles/NOTES_icnn.md). Textually included by csrc/icnn.hip inside its anonymous namespace when the unit is compiled with
// -DDXO_EXPERIMENTS (scripts/exp/build_variant.py); the product build of libdxo_hip.so does not see this file and answers
// ctx option icnn_variant = 3 / 4 with DXO_E_OPTION.
//   icnn_mfma_bf16x3_pipe   (icnn_variant 3)  scalar fp32 arithmetic, phase 1 of the next half-tile issued between the vector
//                                              instructions of this one's phases 2: bit-identical to the default, 3.05 ms against 2.75
//   icnn_mfma_bf16x3_hybrid (icnn_variant 4)  scalar operands inside phase 1's MFMAs, packed phases 2 and 3: bit-identical, 2.93 ms
#pragma once

// ---- the phases of one half-tile (32 points, both half-waves on the same points, each on half of the neurons) as device
// functions with SCALAR fp32 arithmetic, for icnn_mfma_bf16x3_pipe, which runs phase 1 of the NEXT half-tile beside phases 2 of
// this one. The arithmetic (operations and their order per accumulator) is that of icnn_mfma_bf16x3_packed, bit for bit.
struct IcnnTabs {
    const icnn_u32x4* A_l;    // fragment (c, t, st, part) at (((c*2+t)*4+st)*3+part)*64, this lane's 16 bytes
    const icnn_u32x4* AT_l;   // beta GEMM: (t, g, part) at ((t*4+g)*3+part)*64
    const float* P1_h;        // per neuron pair: A1x A1y A1z d1
    const float* P2_h;        // S2x S2y S2z c2 w3/6
    const float* P3_h;        // the six products A1_k A1_l / 6
};

// The pair functions below call `tk.template at<P>()` at points P spread evenly over their vector instructions (one point per
// ~5): the pipelined kernel issues an MFMA of ANOTHER half-tile at (some of) those points and pins the order there, so that
// the in-order issue finds vector work behind every MFMA; the sequential kernel passes IcnnNoTick.
struct IcnnNoTick {
    template <int P> __device__ __forceinline__ void at() {}
};

// softplus3_pk's statements with three tick points (P0, P0 + 1, P0 + 2)
template <int P0, class Tk>
__device__ __forceinline__ void softplus3_tk(icnn_f2 a, icnn_f2& sp, icnn_f2& s1, icnn_f2& s2, Tk& tk) {
    const icnn_f2 am = {fminf(a.x, 80.0f), fminf(a.y, 80.0f)};
    const icnn_f2 t = am * 1.4426950408889634f;
    const icnn_f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    tk.template at<P0>();
    const icnn_f2 ope = e + 1.0f;
    const icnn_f2 r = {__builtin_amdgcn_rcpf(ope.x), __builtin_amdgcn_rcpf(ope.y)};
    tk.template at<P0 + 1>();
    const icnn_f2 l = icnn_f2{__builtin_amdgcn_logf(ope.x), __builtin_amdgcn_logf(ope.y)} * 0.6931471805599453f;
    sp = icnn_f2{fmaxf(l.x, a.x), fmaxf(l.y, a.y)};
    tk.template at<P0 + 2>();
    s1 = e * r;
    s2 = s1 * r;
}

// icnn_split_pair's statements with one tick point in the middle
template <int P0, class Tk>
__device__ __forceinline__ void icnn_split_pair_tk(icnn_f2 v, unsigned& h, unsigned& m, unsigned& l, Tk& tk) {
    const unsigned vx = __float_as_uint(v.x), vy = __float_as_uint(v.y);
    h = __builtin_amdgcn_perm(vy, vx, 0x07060302u);
    const icnn_f2 vh = {__uint_as_float(vx & 0xffff0000u), __uint_as_float(vy & 0xffff0000u)};
    const icnn_f2 r = v - vh;
    tk.template at<P0>();
    const unsigned rx = __float_as_uint(r.x), ry = __float_as_uint(r.y);
    m = __builtin_amdgcn_perm(ry, rx, 0x07060302u);
    const icnn_f2 rh = {__uint_as_float(rx & 0xffff0000u), __uint_as_float(ry & 0xffff0000u)};
    const icnn_f2 q = r - rh;
    l = __builtin_amdgcn_perm(__float_as_uint(q.y), __float_as_uint(q.x), 0x07060302u);
}

// the table rows of a neuron pair: phase 1 (A1x A1y | A1z d1), phase 2 (S2x S2y | S2z c2 | w3 / 6)
struct IcnnRow1 { float4 ta, tb; };
struct IcnnRow2 { float4 ta, tb; float2 tw; };
template <int ST, int K>
__device__ __forceinline__ IcnnRow1 icnn_row1(const IcnnTabs& T) {
    const int pr = icnn_row(ST >> 1, 8 * (ST & 1) + 2 * K) >> 1;
    return {*reinterpret_cast<const float4*>(T.P1_h + pr * 8), *reinterpret_cast<const float4*>(T.P1_h + pr * 8 + 4)};
}
template <int G, int K>
__device__ __forceinline__ IcnnRow2 icnn_row2(const IcnnTabs& T) {
    const int pr = icnn_row(G >> 1, 8 * (G & 1) + 2 * K) >> 1;
    return {*reinterpret_cast<const float4*>(T.P2_h + pr * 12), *reinterpret_cast<const float4*>(T.P2_h + pr * 12 + 4),
            *reinterpret_cast<const float2*>(T.P2_h + pr * 12 + 8)};
}

// phase 1, K-step ST, neuron pair K of this lane's four: the dwords K of the B fragments of h1 = softplus^2 and u = softplus softplus'
// (~56 vector instructions; tick points BASE .. BASE + 7)
template <int ST, int K, int BASE = 0, class Tk = IcnnNoTick>
__device__ __forceinline__ void icnn_p1_pair(const IcnnRow1& row, float xs0, float xs1, float xs2, icnn_f2& cph, IcnnSplitW& Bhw, IcnnSplitW& Buw, Tk&& tk = Tk()) {
    const float4 ta = row.ta, tb = row.tb;   // A1x pair, A1y pair | A1z pair, d1 pair
    const icnn_f2 a = icnn_fma2(icnn_f2{ta.x, ta.y}, icnn_f2{xs0, xs0},
                                icnn_fma2(icnn_f2{ta.z, ta.w}, icnn_f2{xs1, xs1},
                                          icnn_fma2(icnn_f2{tb.x, tb.y}, icnn_f2{xs2, xs2}, icnn_f2{tb.z, tb.w})));
    tk.template at<BASE>();
    icnn_f2 sp, s1, s2;
    softplus3_tk<BASE + 1>(a, sp, s1, s2, tk);
    const icnn_f2 hv = sp * sp, uv = sp * s1;
    cph = icnn_fma2(sp, s2, s1 * s1);   // phi''(a1) * 6: the 1/6 sits in the table of products (phase 3)
    DXO_ICNN_PIN2(cph)
    tk.template at<BASE + 4>();
    icnn_split_pair_tk<BASE + 5>(hv, Bhw.h[K], Bhw.m[K], Bhw.l[K], tk);
    tk.template at<BASE + 6>();
    icnn_split_pair_tk<BASE + 7>(uv, Buw.h[K], Buw.m[K], Buw.l[K], tk);
}

// phase 1, K-step ST: this lane's eight layer-1 neurons (four pairs) -> the B fragments
template <int ST>
__device__ __forceinline__ void icnn_p1_operands(const IcnnTabs& T, float xs0, float xs1, float xs2, icnn_f2 (&cph)[16], IcnnSplit& Bh, IcnnSplit& Bu) {
    IcnnSplitW Bhw, Buw;
    icnn_p1_pair<ST, 0>(icnn_row1<ST, 0>(T), xs0, xs1, xs2, cph[ST * 4 + 0], Bhw, Buw);
    icnn_p1_pair<ST, 1>(icnn_row1<ST, 1>(T), xs0, xs1, xs2, cph[ST * 4 + 1], Bhw, Buw);
    icnn_p1_pair<ST, 2>(icnn_row1<ST, 2>(T), xs0, xs1, xs2, cph[ST * 4 + 2], Bhw, Buw);
    icnn_p1_pair<ST, 3>(icnn_row1<ST, 3>(T), xs0, xs1, xs2, cph[ST * 4 + 3], Bhw, Buw);
    Bh = icnn_pack(Bhw);
    Bu = icnn_pack(Buw);
}

__device__ __forceinline__ void icnn_load_A(const IcnnTabs& T, int st, int c, IcnnSplit (&A)[2]) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        A[jt].h = T.A_l[(((c * 2 + jt) * 4 + st) * 3 + 0) * 64];
        A[jt].m = T.A_l[(((c * 2 + jt) * 4 + st) * 3 + 1) * 64];
        A[jt].l = T.A_l[(((c * 2 + jt) * 4 + st) * 3 + 2) * 64];
    }
}
__device__ __forceinline__ void icnn_load_AT(const IcnnTabs& T, int g, IcnnSplit (&A)[2]) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        A[it].h = T.AT_l[((it * 4 + g) * 3 + 0) * 64];
        A[it].m = T.AT_l[((it * 4 + g) * 3 + 1) * 64];
        A[it].l = T.AT_l[((it * 4 + g) * 3 + 2) * 64];
    }
}

// phase 1, K-step ST: [a2 | g0 | g1 | g2] += [A_0 h1 | A_1 u | A_2 u | A_3 u], 48 MFMAs
template <int ST>
__device__ __forceinline__ void icnn_p1_mfma(const IcnnTabs& T, const IcnnSplit& Bh, const IcnnSplit& Bu, f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        IcnnSplit A[2];
        icnn_load_A(T, ST, c, A);
        icnn_mma6(A, c == 0 ? Bh : Bu, acc[0][c], acc[1][c]);
    }
}

// an accumulator element where the pipelined kernel keeps it (an AGPR: MFMA C/D operands live there when a wave owns all 512
// registers), read at the point of use — left to itself the compiler copies all 128 elements in one block when the half-tile's
// phase 2 begins, 128 vector instructions with no MFMA to hide behind
template <bool AGPR>
__device__ __forceinline__ float icnn_acc_read(const f32x16& c, int q) {
    if constexpr (AGPR) {
        float r;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(c[q]));
        return r;
    } else return c[q];
}

// phase 2, group G, pair K: two layer-2 neurons out of the accumulator registers -> first Hessian term, gradient sums, and the
// deltas as dword K of the B fragment of the beta GEMM (~80 vector instructions; tick points 0 .. 15)
template <int G, int K, class Tk = IcnnNoTick, bool AGPR = false>
__device__ __forceinline__ void icnn_p2_pair(const IcnnRow2& row, float xs0, float xs1, float xs2, const f32x16 (&acc)[2][4], icnn_f2 (&res)[9],
                                             IcnnSplitW& Bdw, Tk&& tk = Tk()) {
    constexpr int jt = G >> 1;
    constexpr int q = 8 * (G & 1) + 2 * K;
    const float4 ta = row.ta, tb = row.tb;   // S2x pair, S2y pair | S2z pair, c2 pair
    const float2 tw = row.tw;                // w3 / 6 pair
    const icnn_f2 S2x = {ta.x, ta.y}, S2y = {ta.z, ta.w}, S2z = {tb.x, tb.y}, w6 = {tw.x, tw.y};
    const icnn_f2 a2 = icnn_f2{icnn_acc_read<AGPR>(acc[jt][0], q), icnn_acc_read<AGPR>(acc[jt][0], q + 1)} +
                       icnn_fma2(S2x, icnn_f2{xs0, xs0}, icnn_fma2(S2y, icnn_f2{xs1, xs1}, icnn_fma2(S2z, icnn_f2{xs2, xs2}, icnn_f2{tb.z, tb.w})));
    tk.template at<0>();
    const icnn_f2 g0 = icnn_f2{icnn_acc_read<AGPR>(acc[jt][1], q), icnn_acc_read<AGPR>(acc[jt][1], q + 1)} + S2x;
    const icnn_f2 g1 = icnn_f2{icnn_acc_read<AGPR>(acc[jt][2], q), icnn_acc_read<AGPR>(acc[jt][2], q + 1)} + S2y;
    tk.template at<1>();
    const icnn_f2 g2 = icnn_f2{icnn_acc_read<AGPR>(acc[jt][3], q), icnn_acc_read<AGPR>(acc[jt][3], q + 1)} + S2z;
    icnn_f2 sp, s1, s2;
    softplus3_tk<2>(a2, sp, s1, s2, tk);
    const icnn_f2 dl = w6 * sp * s1;
    tk.template at<5>();
    const icnn_f2 curv = w6 * icnn_fma2(sp, s2, s1 * s1);
    tk.template at<6>();
    icnn_split_pair_tk<7>(dl, Bdw.h[K], Bdw.m[K], Bdw.l[K], tk);
    tk.template at<8>();
    const icnn_f2 cg0 = curv * g0, cg1 = curv * g1, cg2 = curv * g2;
    tk.template at<9>();
    res[0] = icnn_fma2(dl, g0, res[0]); res[1] = icnn_fma2(dl, g1, res[1]); res[2] = icnn_fma2(dl, g2, res[2]);
    tk.template at<10>();
    res[3] = icnn_fma2(cg0, g0, res[3]); res[4] = icnn_fma2(cg0, g1, res[4]);
    tk.template at<11>();
    res[5] = icnn_fma2(cg0, g2, res[5]);
    res[6] = icnn_fma2(cg1, g1, res[6]);
    tk.template at<12>();
    res[7] = icnn_fma2(cg1, g2, res[7]); res[8] = icnn_fma2(cg2, g2, res[8]);
    tk.template at<13>();
#pragma unroll
    for (int r = 0; r < 9; ++r) DXO_ICNN_PIN2(res[r])
    tk.template at<14>();
    tk.template at<15>();
}

// phase 2, group G: eight layer-2 neurons (four pairs)
template <int G>
__device__ __forceinline__ void icnn_p2_operands(const IcnnTabs& T, float xs0, float xs1, float xs2, const f32x16 (&acc)[2][4], icnn_f2 (&res)[9],
                                                 IcnnSplit& Bd) {
    IcnnSplitW Bdw;
    icnn_p2_pair<G, 0>(icnn_row2<G, 0>(T), xs0, xs1, xs2, acc, res, Bdw);
    icnn_p2_pair<G, 1>(icnn_row2<G, 1>(T), xs0, xs1, xs2, acc, res, Bdw);
    icnn_p2_pair<G, 2>(icnn_row2<G, 2>(T), xs0, xs1, xs2, acc, res, Bdw);
    icnn_p2_pair<G, 3>(icnn_row2<G, 3>(T), xs0, xs1, xs2, acc, res, Bdw);
    Bd = icnn_pack(Bdw);
}

// phase 2, group G: beta += W2p^T delta, 12 MFMAs
template <int G>
__device__ __forceinline__ void icnn_p2_mfma(const IcnnTabs& T, const IcnnSplit& Bd, f32x16 (&bacc)[2]) {
    IcnnSplit A[2];
    icnn_load_AT(T, G, A);
    icnn_mma6(A, Bd, bacc[0], bacc[1]);
}

// phase 3: second Hessian term, beta_i (accumulator registers) x phi''(a1_i) (cph) x A1_i A1_i^T / 6 (table)
__device__ __forceinline__ void icnn_p3(const IcnnTabs& T, const f32x16 (&bacc)[2], const icnn_f2 (&cph)[16], icnn_f2 (&res)[9]) {
#pragma unroll
    for (int ip = 0; ip < 16; ++ip) {
        const int it = ip >> 3, q = 2 * (ip & 7);
        const int pr = icnn_row(it, q) >> 1;
        const float4 pa = *reinterpret_cast<const float4*>(T.P3_h + pr * 12);
        const float4 pb = *reinterpret_cast<const float4*>(T.P3_h + pr * 12 + 4);
        const float4 pc = *reinterpret_cast<const float4*>(T.P3_h + pr * 12 + 8);
        const icnn_f2 c = icnn_f2{bacc[it][q], bacc[it][q + 1]} * cph[ip];
        res[3] = icnn_fma2(c, icnn_f2{pa.x, pa.y}, res[3]); res[4] = icnn_fma2(c, icnn_f2{pa.z, pa.w}, res[4]);
        res[5] = icnn_fma2(c, icnn_f2{pb.x, pb.y}, res[5]); res[6] = icnn_fma2(c, icnn_f2{pb.z, pb.w}, res[6]);
        res[7] = icnn_fma2(c, icnn_f2{pc.x, pc.y}, res[7]); res[8] = icnn_fma2(c, icnn_f2{pc.z, pc.w}, res[8]);
#pragma unroll
        for (int r = 3; r < 9; ++r) DXO_ICNN_PIN2(res[r])
    }
}

// both half-waves hold partial sums of the SAME 32 points: add them, the half-wave that owns the points keeps them (LDS slot)
__device__ __forceinline__ void icnn_reduce_store(const icnn_f2 (&res)[9], bool own, float* minep) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float part = res[q].x + res[q].y;
        const float tot = part + xor32(part);
        if (own) minep[q * 64] = tot;
    }
}

__device__ __forceinline__ void icnn_zero(f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[jt][c][q] = 0.f;
}

// the fp32 features of a point (demo_hyperelasticity.py:263-286) from its deformation gradient
__device__ __forceinline__ void icnn_features(const dxo_f64x2& f01, const dxo_f64x2& f23, float& x0, float& x1, float& x2) {
    const double Fv[4] = {f01.x, f01.y, f23.x, f23.y};
    const double t = Fv[0] * Fv[0] + Fv[1] * Fv[1] + Fv[2] * Fv[2] + Fv[3] * Fv[3];
    const double D = Fv[0] * Fv[3] - Fv[1] * Fv[2];
    const double aD = fabs(D);
    const double m = hyper_pow_m23(aD), nn = m * m;
    x0 = (float)((t + 1.0) * m - 3.0); x1 = (float)((t + D * D) * nn - 3.0); x2 = (float)((aD - 1.0) * (aD - 1.0));
}

// after both half-tiles: the fp64 feature derivatives of the lane's own point, chain rule, stores
__device__ __forceinline__ void icnn_finish(const dxo_f64x2& f01, const dxo_f64x2& f23, const float* minep, const IcnnSmall<float>& small,
                                            double* __restrict__ dP, double* __restrict__ P, int64_t pidx) {
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

// A fragments and the pair tables of the split-bf16 kernels, filled once per workgroup
template <int BLOCK>
__device__ __forceinline__ void icnn_fill_lds(const float* __restrict__ wT1, const float* __restrict__ wW2, const float* __restrict__ wT2,
                                              unsigned short* sA3, unsigned short* sAT3, float* sP1, float* sP3, float* sP2) {
    constexpr int FRAG = 64 * 8;   // bf16 per fragment (64 lanes x 8)
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
}

// ------------------------------------------------------------------ SCALAR fp32 vector arithmetic, software-pipelined over half-tiles (icnn_variant = 3)
// profiles/r05_mfma_gap_probe.txt (hand-placed streams, s_memtime): a v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32
// cycles and up to SIX plain vector instructions issued behind it are free (37 cycles per MFMA with six v_fma_f32, 41 with
// seven, +4 each beyond; a transcendental counts double; one packed-fp32 instruction costs +20) — with one wave per SIMD or
// two. A wave issues in order, so the vector work has to sit BETWEEN the MFMAs in the instruction stream: icnn_mfma_bf16x3
// above runs [200 vector instructions][48 MFMAs] per K-step, and its half-tile costs the sum of both (two waves per SIMD run in
// lock step: started half a period apart they measure the same, scripts/exp/archive/icnn_skew.py). Here ONE wave per SIMD (512
// registers: two sets of accumulators) runs phase 1 of half-tile k + 1 (192 MFMAs) in the same instruction stream as phases 2
// of half-tile k (1 300 vector instructions + 48 MFMAs), slice by slice (K-step st beside group g = st), and the operands of
// K-step st + 1 are formed beside the MFMAs of K-step st.
struct IcnnHalf {     // what a half-tile carries from its phase 1 to its phases 2 and 3
    float xs0, xs1, xs2;
    bool own;
    icnn_f2 cph[16];
    f32x16 acc[2][4];
};

// what the NEXT phase 1's K-step 0 needs and produces while the current half-tiles still own H0 / H1: formed in the last slice
// of a slot, moved into the half-tile's record when the slot is over
struct IcnnStage {
    float xs0, xs1, xs2;
    icnn_f2 cph0[4];
};

// MFMA J (0..11) of icnn_mma6's sequence; FIRST: the accumulators start here (C = 0 as the instruction's inline constant
// instead of 128 v_accvgpr_write per half-tile; 0 + x is x, the bits do not change)
template <int J, bool FIRST = false>
__device__ __forceinline__ void icnn_mma6_j(const IcnnSplit (&A)[2], const IcnnSplit& B, f32x16& c0, f32x16& c1) {
    constexpr int jt = J & 1, step = J >> 1;
    f32x16& c = jt ? c1 : c0;
    if constexpr (step == 0) {
        if constexpr (FIRST) {
            f32x16 z;
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = 0.f;
            c = icnn_mfma_bf16(A[jt].l, B.h, z);
        } else c = icnn_mfma_bf16(A[jt].l, B.h, c);
    }
    else if constexpr (step == 1) c = icnn_mfma_bf16(A[jt].h, B.l, c);
    else if constexpr (step == 2) c = icnn_mfma_bf16(A[jt].m, B.m, c);
    else if constexpr (step == 3) c = icnn_mfma_bf16(A[jt].m, B.h, c);
    else if constexpr (step == 4) c = icnn_mfma_bf16(A[jt].h, B.m, c);
    else c = icnn_mfma_bf16(A[jt].h, B.h, c);
}

// nothing but scalar instructions and global loads may cross a tick: vector instructions, MFMAs AND the LDS reads stay where
// the source puts them (a single wave per SIMD has nobody to hide an LDS latency behind: every read is requested a fixed
// number of points before its first use)
#define DXO_ICNN_TICK_FENCE __builtin_amdgcn_sched_barrier(0x0004 | 0x0020);

// what one chunk hands to the next: the fragments and table rows requested while it ran
struct IcnnAhead {
    IcnnSplit A[2];     // fragments of the next chunk's matrix
    IcnnRow2 r2;        // table row of the next chunk's phase-2 pair
};

// The ticks of chunk (S, K): 24 points over pair K of cur's group S (points 0..15) and pair K of nxt's next operands (16..23).
//   even point 2 j:  MFMA j of nxt's K-step S, matrix K   (fragments `now.A`, requested during the previous chunk)
//   odd point 2 j + 1, chunk 0 of slices 1..3 only:  MFMA j of the beta GEMM of cur's group S - 1 (fragments AT, deltas Bd)
//   point 8: request this chunk's phase-1 table row; point 14: the next chunk's fragments; point 18: its phase-2 table row;
//   point 20 of chunk 3: the beta fragments of this slice's group
template <int S, int K>
struct IcnnTick {
    const IcnnTabs& T;
    const IcnnAhead& now;
    IcnnAhead& nxt;
    IcnnRow1& r1;
    IcnnSplit (&AT)[2];
    const IcnnSplit& B;
    f32x16& c0;
    f32x16& c1;
    const IcnnSplit& Bd;
    f32x16 (&bacc)[2];
    template <int P>
    __device__ __forceinline__ void at() {
        if constexpr (P % 2 == 0) icnn_mma6_j<P / 2, S == 0>(now.A, B, c0, c1);
        else if constexpr (K == 0 && S > 0) icnn_mma6_j<(P - 1) / 2, S == 1>(AT, Bd, bacc[0], bacc[1]);
        constexpr int S1 = S < 3 ? S + 1 : 0;
        if constexpr (P == 8) r1 = icnn_row1<S1, K>(T);
        if constexpr (P == 14) icnn_load_A(T, K < 3 ? S : S1, (K + 1) & 3, nxt.A);
        if constexpr (P == 18) nxt.r2 = icnn_row2<(K < 3 ? S : S1), (K + 1) & 3>(T);
        if constexpr (P == 20 && K == 3 && S < 3) icnn_load_AT(T, S, AT);
        DXO_ICNN_TICK_FENCE
    }
};

// One slice = K-step S of phase 1 of half-tile `nxt` beside group S of phase 2 of half-tile `cur`, chunk K = 0..3:
//   the 12 MFMAs of matrix K  |  pair K of cur's group S (~80 vector instructions)  |  pair K of the operands of nxt's K-step
//   S + 1 (~56) — for S = 3 of K-step 0 of the half-tile AFTER nxt (`stg`)  |  chunk 0, S > 0: the 12 beta MFMAs of group S - 1.
// Bh, Bu hold nxt's K-step S operands on entry and the following K-step's on exit; Bd cur's group S - 1 deltas on entry and group
// S's on exit; a0 what chunk 0 needs on entry and what the next slice's chunk 0 needs on exit.
template <int S>
__device__ __forceinline__ void icnn_slice(const IcnnTabs& T, IcnnHalf& cur, IcnnHalf& nxt, IcnnStage& stg, icnn_f2 (&res)[9], f32x16 (&bacc)[2],
                                           IcnnSplit& Bh, IcnnSplit& Bu, IcnnSplit& Bd, IcnnAhead& a0, IcnnAhead& a1, IcnnSplit (&AT)[2]) {
    IcnnSplitW Bdw, Bhw, Buw;
    IcnnRow1 r1;
    constexpr int S1 = S < 3 ? S + 1 : 0;
#define DXO_ICNN_CHUNK(K_, NOW_, NXT_)                                                                                             \
    {                                                                                                                               \
        IcnnTick<S, K_> tk{T, NOW_, NXT_, r1, AT, K_ == 0 ? Bh : Bu, nxt.acc[0][K_], nxt.acc[1][K_], Bd, bacc};                      \
        icnn_p2_pair<S, K_, IcnnTick<S, K_>&, DXO_ICNN_PIPE_AGPR>(NOW_.r2, cur.xs0, cur.xs1, cur.xs2, cur.acc, res, Bdw, tk);        \
        if (S < 3) icnn_p1_pair<S1, K_, 16>(r1, nxt.xs0, nxt.xs1, nxt.xs2, nxt.cph[S1 * 4 + K_], Bhw, Buw, tk);                      \
        else icnn_p1_pair<0, K_, 16>(r1, stg.xs0, stg.xs1, stg.xs2, stg.cph0[K_], Bhw, Buw, tk);                                     \
    }
    DXO_ICNN_CHUNK(0, a0, a1)
    DXO_ICNN_CHUNK(1, a1, a0)
    DXO_ICNN_CHUNK(2, a0, a1)
    DXO_ICNN_CHUNK(3, a1, a0)
#undef DXO_ICNN_CHUNK
    Bd = icnn_pack(Bdw);
    Bh = icnn_pack(Bhw);
    Bu = icnn_pack(Buw);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void icnn_mfma_bf16x3_pipe(const float* __restrict__ wT1, const float* __restrict__ wW2,
                                                                      const float* __restrict__ wT2, IcnnSmall<float> small, int64_t n,
                                                                      const double* __restrict__ F, double* __restrict__ dP,
                                                                      double* __restrict__ P) {
    constexpr int BLOCK = WAVES * 64;
    constexpr int FRAG = 64 * 8;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) unsigned short sA3[4 * 2 * 4 * 3 * FRAG];
    __shared__ __attribute__((aligned(16))) unsigned short sAT3[2 * 4 * 3 * FRAG];
    __shared__ __attribute__((aligned(16))) float sP1[32 * 8];
    __shared__ __attribute__((aligned(16))) float sP3[32 * 12];
    __shared__ __attribute__((aligned(16))) float sP2[32 * 12];
    __shared__ float sMine[WAVES * 9 * 64];
    float* minep = sMine + wave * (9 * 64) + lane;
    icnn_fill_lds<BLOCK>(wT1, wW2, wT2, sA3, sAT3, sP1, sP3, sP2);
    __syncthreads();
    const IcnnTabs T = {reinterpret_cast<const icnn_u32x4*>(sA3) + lane, reinterpret_cast<const icnn_u32x4*>(sAT3) + lane, sP1 + 2 * h * 8,
                        sP2 + 2 * h * 12, sP3 + 2 * h * 12};

    const int64_t n_tiles = (n + 63) / 64;
    const int64_t tile_step = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile >= n_tiles) return;
    auto load_F = [&](int64_t tl, dxo_f64x2& a, dxo_f64x2& b) {   // a tile past the end repeats the last one (never stored)
        const int64_t tc = tl < n_tiles ? tl : n_tiles - 1;
        const int64_t pl = tc * 64 + lane < n ? tc * 64 + lane : n - 1;
        a = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[0];
        b = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[1];
    };
    dxo_f64x2 f01, f23, f01n, f23n;
    load_F(tile, f01, f23);
    load_F(tile + tile_step, f01n, f23n);
    float x0, x1, x2;
    icnn_features(f01, f23, x0, x1, x2);
    IcnnHalf H0, H1;     // H0: half-tile tt = 0 of the current tile (then of the next one), H1: tt = 1
    IcnnStage stg;
    IcnnSplit Bh, Bu, Bd;
    IcnnAhead a0, a1;
    IcnnSplit AT[2];
    // prologue: phase 1 of (tile, tt = 0) with nothing beside it, and K-step 0's operands of (tile, tt = 1)
    float xp0 = xor32(x0), xp1 = xor32(x1), xp2 = xor32(x2);   // the shuffles run with every lane active, the selects follow
    H0.own = (h == 0);
    H0.xs0 = H0.own ? x0 : xp0; H0.xs1 = H0.own ? x1 : xp1; H0.xs2 = H0.own ? x2 : xp2;
    icnn_zero(H0.acc);
    icnn_p1_operands<0>(T, H0.xs0, H0.xs1, H0.xs2, H0.cph, Bh, Bu); icnn_p1_mfma<0>(T, Bh, Bu, H0.acc);
    icnn_p1_operands<1>(T, H0.xs0, H0.xs1, H0.xs2, H0.cph, Bh, Bu); icnn_p1_mfma<1>(T, Bh, Bu, H0.acc);
    icnn_p1_operands<2>(T, H0.xs0, H0.xs1, H0.xs2, H0.cph, Bh, Bu); icnn_p1_mfma<2>(T, Bh, Bu, H0.acc);
    icnn_p1_operands<3>(T, H0.xs0, H0.xs1, H0.xs2, H0.cph, Bh, Bu); icnn_p1_mfma<3>(T, Bh, Bu, H0.acc);
    H1.own = (h == 1);
    H1.xs0 = H1.own ? x0 : xp0; H1.xs1 = H1.own ? x1 : xp1; H1.xs2 = H1.own ? x2 : xp2;
    icnn_p1_operands<0>(T, H1.xs0, H1.xs1, H1.xs2, H1.cph, Bh, Bu);
    icnn_load_A(T, 0, 0, a0.A);
    a0.r2 = icnn_row2<0, 0>(T);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t pidx = tile * 64 + lane;
        asm volatile("" ::: "memory");   // keep the loop-invariant LDS reads inside the loop
        icnn_f2 res[9];
        f32x16 bacc[2];
        // the next tile's features (its F was requested a tile ago); request the one after
        const dxo_f64x2 f01c = f01, f23c = f23;
        f01 = f01n; f23 = f23n;
        load_F(tile + 2 * tile_step, f01n, f23n);
        icnn_features(f01, f23, x0, x1, x2);
        xp0 = xor32(x0); xp1 = xor32(x1); xp2 = xor32(x2);
        // ---- slot A: phases 2 of (tile, 0) beside phase 1 of (tile, 1); its last slice forms K-step 0 of (tile + step, 0)
        stg.xs0 = (h == 0) ? x0 : xp0; stg.xs1 = (h == 0) ? x1 : xp1; stg.xs2 = (h == 0) ? x2 : xp2;
#pragma unroll
        for (int q = 0; q < 9; ++q) res[q] = icnn_f2{0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        icnn_slice<0>(T, H0, H1, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<1>(T, H0, H1, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<2>(T, H0, H1, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<3>(T, H0, H1, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        __builtin_amdgcn_sched_barrier(0);
        icnn_p2_mfma<3>(T, Bd, bacc);      // the last group's beta MFMAs and phase 3 have nothing beside them
        icnn_p3(T, bacc, H0.cph, res);
        icnn_reduce_store(res, H0.own, minep);
        // H0 becomes (tile + step, 0): what the last slice staged
        H0.xs0 = stg.xs0; H0.xs1 = stg.xs1; H0.xs2 = stg.xs2;
#pragma unroll
        for (int k = 0; k < 4; ++k) H0.cph[k] = stg.cph0[k];
        // ---- slot B: phases 2 of (tile, 1) beside phase 1 of (tile + step, 0); its last slice forms K-step 0 of (tile + step, 1)
        stg.xs0 = (h == 1) ? x0 : xp0; stg.xs1 = (h == 1) ? x1 : xp1; stg.xs2 = (h == 1) ? x2 : xp2;
#pragma unroll
        for (int q = 0; q < 9; ++q) res[q] = icnn_f2{0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        icnn_slice<0>(T, H1, H0, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<1>(T, H1, H0, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<2>(T, H1, H0, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        icnn_slice<3>(T, H1, H0, stg, res, bacc, Bh, Bu, Bd, a0, a1, AT);
        __builtin_amdgcn_sched_barrier(0);
        icnn_p2_mfma<3>(T, Bd, bacc);
        icnn_p3(T, bacc, H1.cph, res);
        icnn_reduce_store(res, H1.own, minep);
        H1.xs0 = stg.xs0; H1.xs1 = stg.xs1; H1.xs2 = stg.xs2;
#pragma unroll
        for (int k = 0; k < 4; ++k) H1.cph[k] = stg.cph0[k];
        asm volatile("" ::: "memory");
        if (pidx < n) icnn_finish(f01c, f23c, minep, small, dP, P, pidx);
    }
}

// ------------------------------------------------------------------ HYBRID (icnn_variant = 4): scalar operands inside the MFMAs of phase 1, packed phases 2 and 3
// What the gap probe allows (profiles/r05_mfma_gap_probe.txt) and what the two kernels above each get half of: plain vector
// instructions ride free behind an MFMA (<= 6 per 32 cycles), packed ones never do but need ~40 % fewer instructions. Per half-tile:
//   phase 1  192 MFMAs; the ~900 SCALAR instructions that form the operands of the NEXT K-step (for the last K-step: K-step 0 of the
//            next half-tile) sit at tick points between them, ~4.7 per MFMA — hidden; fragments and table rows requested a chunk ahead
//   phases 2, 3  the default kernel's PACKED statements, with no MFMA in flight except the 12 beta MFMAs that end each group
// One wave per SIMD (a packed instruction of a second wave would wait behind this wave's MFMAs). Same operations in the same order
// per accumulator as the other two kernels: bit-identical outputs.
template <int ST, int K>
struct IcnnTickH {      // chunk (ST, K) of phase 1: the 12 MFMAs of K-step ST, matrix K, over the 8 tick points of one scalar pair
    const IcnnTabs& T;
    const IcnnSplit (&A)[2];
    IcnnSplit (&An)[2];
    IcnnRow1& r1n;
    const IcnnSplit& B;
    f32x16& c0;
    f32x16& c1;
    template <int P>
    __device__ __forceinline__ void at() {
        constexpr int j0 = (P / 2) * 3 + (P % 2) * 2;        // MFMAs per point: 2, 1, 2, 1, 2, 1, 2, 1
        icnn_mma6_j<j0, ST == 0>(A, B, c0, c1);
        if constexpr (P % 2 == 0) icnn_mma6_j<j0 + 1, ST == 0>(A, B, c0, c1);
        // the next chunk: (ST, K + 1) or (ST + 1, 0); its vector work is pair K' of K-step ST' + 1 (of the next half-tile's K-step 0 for ST' = 3)
        constexpr int STn = K < 3 ? ST : ST + 1, Kn = (K + 1) & 3;
        if constexpr (STn < 4) {
            if constexpr (P == 2) r1n = icnn_row1<(STn < 3 ? STn + 1 : 0), Kn>(T);
            if constexpr (P == 4) icnn_load_A(T, STn, Kn, An);
        }
        DXO_ICNN_TICK_FENCE
    }
};

// phases 2 and 3 of a half-tile with packed fp32 arithmetic: the statements of icnn_mfma_bf16x3_packed
__device__ __forceinline__ void icnn_p23_packed(const IcnnTabs& T, float xs0, float xs1, float xs2, const f32x16 (&acc)[2][4], const icnn_f2 (&cph)[16],
                                                icnn_f2p (&res)[9]) {
#pragma unroll
    for (int q = 0; q < 9; ++q) res[q] = icnn_f2p{0.f, 0.f};
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
            const float4 ta = *reinterpret_cast<const float4*>(T.P2_h + pr * 12);       // S2x pair, S2y pair
            const float4 tb = *reinterpret_cast<const float4*>(T.P2_h + pr * 12 + 4);   // S2z pair, c2 pair
            const float2 tw = *reinterpret_cast<const float2*>(T.P2_h + pr * 12 + 8);   // w3 / 6 pair
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
        icnn_load_AT(T, g, A);
        icnn_mma6(A, icnn_pack(Bdw), bacc[0], bacc[1]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ip = 0; ip < 16; ++ip) {
        const int it = ip >> 3, q = 2 * (ip & 7);
        const int pr = icnn_row(it, q) >> 1;
        const float4 pa = *reinterpret_cast<const float4*>(T.P3_h + pr * 12);
        const float4 pb = *reinterpret_cast<const float4*>(T.P3_h + pr * 12 + 4);
        const float4 pc = *reinterpret_cast<const float4*>(T.P3_h + pr * 12 + 8);
        const icnn_f2p c = icnn_f2p{bacc[it][q], bacc[it][q + 1]} * icnn_f2p{cph[ip].x, cph[ip].y};
        res[3] = icnn_fma2_p(c, icnn_f2p{pa.x, pa.y}, res[3]); res[4] = icnn_fma2_p(c, icnn_f2p{pa.z, pa.w}, res[4]);
        res[5] = icnn_fma2_p(c, icnn_f2p{pb.x, pb.y}, res[5]); res[6] = icnn_fma2_p(c, icnn_f2p{pb.z, pb.w}, res[6]);
        res[7] = icnn_fma2_p(c, icnn_f2p{pc.x, pc.y}, res[7]); res[8] = icnn_fma2_p(c, icnn_f2p{pc.z, pc.w}, res[8]);
#pragma unroll
        for (int r = 3; r < 9; ++r) DXO_ICNN_PINP(res[r])
    }
}

// K-step ST of phase 1 in four chunks: matrix K's 12 MFMAs inside pair K of the next operands (VST = the K-step whose operands these
// are; `cphn` where their phi'' go). Bh, Bu: this K-step's operands on entry, the next one's on exit.
template <int ST>
__device__ __forceinline__ void icnn_hybrid_step(const IcnnTabs& T, float nx0, float nx1, float nx2, icnn_f2* cphn, f32x16 (&acc)[2][4], IcnnSplit& Bh,
                                                 IcnnSplit& Bu, IcnnSplit (&A0)[2], IcnnSplit (&A1)[2], IcnnRow1& r1a, IcnnRow1& r1b) {
    IcnnSplitW Bhw, Buw;
    constexpr int VST = ST < 3 ? ST + 1 : 0;
#define DXO_ICNN_HCHUNK(K_, ACUR_, ANXT_, RCUR_, RNXT_)                                                                           \
    {                                                                                                                               \
        IcnnTickH<ST, K_> tk{T, ACUR_, ANXT_, RNXT_, K_ == 0 ? Bh : Bu, acc[0][K_], acc[1][K_]};                                     \
        icnn_p1_pair<VST, K_, 0>(RCUR_, nx0, nx1, nx2, cphn[K_], Bhw, Buw, tk);                                                      \
    }
    DXO_ICNN_HCHUNK(0, A0, A1, r1a, r1b)
    DXO_ICNN_HCHUNK(1, A1, A0, r1b, r1a)
    DXO_ICNN_HCHUNK(2, A0, A1, r1a, r1b)
    DXO_ICNN_HCHUNK(3, A1, A0, r1b, r1a)
#undef DXO_ICNN_HCHUNK
    Bh = icnn_pack(Bhw);
    Bu = icnn_pack(Buw);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void icnn_mfma_bf16x3_hybrid(const float* __restrict__ wT1, const float* __restrict__ wW2,
                                                                        const float* __restrict__ wT2, IcnnSmall<float> small, int64_t n,
                                                                        const double* __restrict__ F, double* __restrict__ dP,
                                                                        double* __restrict__ P) {
    constexpr int BLOCK = WAVES * 64;
    constexpr int FRAG = 64 * 8;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) unsigned short sA3[4 * 2 * 4 * 3 * FRAG];
    __shared__ __attribute__((aligned(16))) unsigned short sAT3[2 * 4 * 3 * FRAG];
    __shared__ __attribute__((aligned(16))) float sP1[32 * 8];
    __shared__ __attribute__((aligned(16))) float sP3[32 * 12];
    __shared__ __attribute__((aligned(16))) float sP2[32 * 12];
    __shared__ float sMine[WAVES * 9 * 64];
    float* minep = sMine + wave * (9 * 64) + lane;
    icnn_fill_lds<BLOCK>(wT1, wW2, wT2, sA3, sAT3, sP1, sP3, sP2);
    __syncthreads();
    const IcnnTabs T = {reinterpret_cast<const icnn_u32x4*>(sA3) + lane, reinterpret_cast<const icnn_u32x4*>(sAT3) + lane, sP1 + 2 * h * 8,
                        sP2 + 2 * h * 12, sP3 + 2 * h * 12};
    const int64_t n_tiles = (n + 63) / 64;
    const int64_t tile_step = (int64_t)gridDim.x * WAVES;
    int64_t tile = (int64_t)blockIdx.x * WAVES + wave;
    if (tile >= n_tiles) return;
    auto load_F = [&](int64_t tl, dxo_f64x2& a, dxo_f64x2& b) {   // a tile past the end repeats the last one (never stored)
        const int64_t tc = tl < n_tiles ? tl : n_tiles - 1;
        const int64_t pl = tc * 64 + lane < n ? tc * 64 + lane : n - 1;
        a = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[0];
        b = reinterpret_cast<const dxo_f64x2*>(F + pl * 4)[1];
    };
    dxo_f64x2 f01, f23, f01n, f23n;
    load_F(tile, f01, f23);
    load_F(tile + tile_step, f01n, f23n);
    float x0, x1, x2;
    icnn_features(f01, f23, x0, x1, x2);
    float xp0 = xor32(x0), xp1 = xor32(x1), xp2 = xor32(x2);
    // the half-tile in flight: its inputs, its phi'' and its accumulators; `nxt*`: what the last K-step stages for the one after it
    float xs0 = (h == 0) ? x0 : xp0, xs1 = (h == 0) ? x1 : xp1, xs2 = (h == 0) ? x2 : xp2;
    icnn_f2 cph[16], cph0n[4];
    f32x16 acc[2][4];
    IcnnSplit Bh, Bu, A0[2], A1[2];
    IcnnRow1 r1a, r1b;
    icnn_p1_operands<0>(T, xs0, xs1, xs2, cph, Bh, Bu);       // prologue: K-step 0 of (tile, 0), nothing to hide behind
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t pidx = tile * 64 + lane;
        asm volatile("" ::: "memory");   // keep the loop-invariant LDS reads inside the loop
        const dxo_f64x2 f01c = f01, f23c = f23;
        // the next tile's features (its F was requested a tile ago); request the one after
        float y0, y1, y2;
        f01 = f01n; f23 = f23n;
        load_F(tile + 2 * tile_step, f01n, f23n);
        icnn_features(f01, f23, y0, y1, y2);
        const float yp0 = xor32(y0), yp1 = xor32(y1), yp2 = xor32(y2);
#pragma unroll 1
        for (int tt = 0; tt < 2; ++tt) {
            const bool own = (tt == h);
            // the half-tile after this one: (tile, 1) after (tile, 0), (tile + step, 0) after (tile, 1)
            const bool nown = tt == 0 ? (h == 1) : (h == 0);
            const float bx0 = tt == 0 ? x0 : y0, bx1 = tt == 0 ? x1 : y1, bx2 = tt == 0 ? x2 : y2;
            const float bp0 = tt == 0 ? xp0 : yp0, bp1 = tt == 0 ? xp1 : yp1, bp2 = tt == 0 ? xp2 : yp2;
            const float nx0 = nown ? bx0 : bp0, nx1 = nown ? bx1 : bp1, nx2 = nown ? bx2 : bp2;
            // ---- phase 1
            icnn_load_A(T, 0, 0, A0);
            r1a = icnn_row1<1, 0>(T);
            __builtin_amdgcn_sched_barrier(0);
            icnn_hybrid_step<0>(T, xs0, xs1, xs2, cph + 4, acc, Bh, Bu, A0, A1, r1a, r1b);
            icnn_hybrid_step<1>(T, xs0, xs1, xs2, cph + 8, acc, Bh, Bu, A0, A1, r1a, r1b);
            icnn_hybrid_step<2>(T, xs0, xs1, xs2, cph + 12, acc, Bh, Bu, A0, A1, r1a, r1b);
            icnn_hybrid_step<3>(T, nx0, nx1, nx2, cph0n, acc, Bh, Bu, A0, A1, r1a, r1b);
            __builtin_amdgcn_sched_barrier(0);
            // ---- phases 2 and 3, packed
            icnn_f2p res[9];
            icnn_p23_packed(T, xs0, xs1, xs2, acc, cph, res);
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const float part = res[q].x + res[q].y;
                const float tot = part + xor32(part);
                if (own) minep[q * 64] = tot;
            }
            // the staged half-tile becomes the current one
            xs0 = nx0; xs1 = nx1; xs2 = nx2;
#pragma unroll
            for (int k = 0; k < 4; ++k) cph[k] = cph0n[k];
            __builtin_amdgcn_sched_barrier(0);
        }
        x0 = y0; x1 = y1; x2 = y2; xp0 = yp0; xp1 = yp1; xp2 = yp2;
        asm volatile("" ::: "memory");
        if (pidx < n) icnn_finish(f01c, f23c, minep, small, dP, P, pidx);
    }
}

