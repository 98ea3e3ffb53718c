// vm_exp.hip — standalone A/B harness for von Mises d=6 kernel structure experiments (not product code).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/exp/archive/vm_exp.hip -o scripts/exp/vm_exp
// run  : scripts/exp/vm_exp [n_points] [rounds] [launches]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct VmConst { double lmbda, mu2, mu3, sigma_0, H, mu3_H, ratio; };

template <int D, bool RECIP>
__device__ __forceinline__ void vm_return_map(const VmConst& c, const double (&deps)[D], const double (&sn)[D], double p,
                                              double (&sig)[D], double& dp, double (&nrm)[D], double& a, double& b) {
    const double tr_e = deps[0] + deps[1] + deps[2];
    double se[D];
#pragma unroll
    for (int i = 0; i < D; ++i) se[i] = sn[i] + ((i < 3 ? c.lmbda * tr_e : 0.0) + c.mu2 * deps[i]);
    const double mean = (se[0] + se[1] + se[2]) * (1.0 / 3.0);
    double s[D];
#pragma unroll
    for (int i = 0; i < D; ++i) s[i] = i < 3 ? se[i] - mean : se[i];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) ss += s[i] * s[i];
    const double sigma_eq = sqrt(3.0 / 2.0 * ss);
    const double f_el = sigma_eq - c.sigma_0 - c.H * p;
    const double f_plus = (f_el + sqrt(f_el * f_el)) / 2.0;
    dp = f_plus / c.mu3_H;
    if constexpr (RECIP) {
        const double inv = 1.0 / sigma_eq;
        const double beta = c.mu3 * dp * inv;
        const double w = inv * (f_plus / f_el);
#pragma unroll
        for (int i = 0; i < D; ++i) { nrm[i] = s[i] * w; sig[i] = se[i] - beta * s[i]; }
        a = c.mu3 * (c.ratio - beta);
        b = c.mu2 * beta;
    } else {
        const double beta = c.mu3 * dp / sigma_eq;
#pragma unroll
        for (int i = 0; i < D; ++i) { nrm[i] = s[i] / sigma_eq * f_plus / f_el; sig[i] = se[i] - beta * s[i]; }
        a = c.mu3 * (c.ratio - beta);
        b = c.mu2 * beta;
    }
}
__device__ __forceinline__ double c_elas_ij(const VmConst& c, int i, int j) { return ((i < 3 && j < 3) ? c.lmbda : 0.0) + (i == j ? c.mu2 : 0.0); }
__device__ __forceinline__ double dev_ij(int i, int j) { return (i == j ? 1.0 : 0.0) - ((i < 3 && j < 3) ? 1.0 / 3.0 : 0.0); }
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Policy knobs
//   RECIP   : reciprocal instead of 12 divisions
//   LDSMODE : 0 = X(3K)+Y(4K) as shipped; 1 = single 4K region, inputs staged one after the other, sigma stored strided
//   BLOCKED : tile assignment: false = grid-stride over tiles; true = each wave owns a contiguous run of tiles
//   MINW    : launch-bounds waves/SIMD
template <bool RECIP, int LDSMODE, bool BLOCKED, int MINW, bool NT>
__global__ __launch_bounds__(256, MINW) void vm6(VmConst c, long n, const double* __restrict__ deps, const double* __restrict__ sigma_n,
                                                 const double* __restrict__ p, double* __restrict__ C_tang, double* __restrict__ sigma,
                                                 double* __restrict__ dp_out) {
    constexpr int D = 6, CH_VEC = 3, CH_CT = 18, ST = 8, PTS = 64, WAVES = 4;
    constexpr int WAVE_DOUBLES = LDSMODE == 0 ? (PTS * D + PTS * ST) : (PTS * ST);
    __shared__ __attribute__((aligned(16))) double lds[WAVES * WAVE_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* X = lds + wave * WAVE_DOUBLES;
    double* Y = LDSMODE == 0 ? X + PTS * D : X;
    f64x2* X2 = (f64x2*)X;
    f64x2* Y2 = (f64x2*)Y;
    const long n_tiles = n / PTS;  // harness: n multiple of 64
    long t_begin, t_end, t_step;
    const long gw = (long)blockIdx.x * WAVES + wave, nw = (long)gridDim.x * WAVES;
    if (BLOCKED) { const long per = (n_tiles + nw - 1) / nw; t_begin = gw * per; t_end = min(t_begin + per, n_tiles); t_step = 1; }
    else { t_begin = gw; t_end = n_tiles; t_step = nw; }
    for (long tile = t_begin; tile < t_end; tile += t_step) {
        const long p0 = tile * PTS;
        const f64x2* g_e = (const f64x2*)(deps + p0 * D);
        const f64x2* g_s = (const f64x2*)(sigma_n + p0 * D);
        f64x2 ve[CH_VEC], vs[CH_VEC];
#pragma unroll
        for (int k = 0; k < CH_VEC; ++k) { ve[k] = g_e[k * 64 + lane]; vs[k] = g_s[k * 64 + lane]; }
        const double p_l = p[p0 + lane];
        double e[D], sn[D];
        if constexpr (LDSMODE == 0) {
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) { X2[k * 64 + lane] = ve[k]; Y2[k * 64 + lane] = vs[k]; }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) {
                const f64x2 a2 = X2[lane * CH_VEC + k], b2 = Y2[lane * CH_VEC + k];
                e[2 * k] = a2.x; e[2 * k + 1] = a2.y; sn[2 * k] = b2.x; sn[2 * k + 1] = b2.y;
            }
            wave_lds_fence();
        } else {
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) X2[k * 64 + lane] = ve[k];
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) { const f64x2 a2 = X2[lane * CH_VEC + k]; e[2 * k] = a2.x; e[2 * k + 1] = a2.y; }
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) X2[k * 64 + lane] = vs[k];
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) { const f64x2 b2 = X2[lane * CH_VEC + k]; sn[2 * k] = b2.x; sn[2 * k + 1] = b2.y; }
            wave_lds_fence();
        }
        double sig[D], nrm[D], dp, a, b;
        vm_return_map<D, RECIP>(c, e, sn, p_l, sig, dp, nrm, a, b);
        if constexpr (LDSMODE == 0) {
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) {
                X2[lane * CH_VEC + k] = f64x2{sig[2 * k], sig[2 * k + 1]};
                Y2[lane * (ST / 2) + k] = f64x2{nrm[2 * k], nrm[2 * k + 1]};
            }
            Y2[lane * (ST / 2) + CH_VEC] = f64x2{a, b};
            wave_lds_fence();
            f64x2* g_o = (f64x2*)(sigma + p0 * D);
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) { if (NT) __builtin_nontemporal_store(X2[k * 64 + lane], g_o + k * 64 + lane); else g_o[k * 64 + lane] = X2[k * 64 + lane]; }
        } else {
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) Y2[lane * (ST / 2) + k] = f64x2{nrm[2 * k], nrm[2 * k + 1]};
            Y2[lane * (ST / 2) + CH_VEC] = f64x2{a, b};
            wave_lds_fence();
            f64x2* g_o = (f64x2*)(sigma + (p0 + lane) * D);  // strided per-lane 48 B
#pragma unroll
            for (int k = 0; k < CH_VEC; ++k) { const f64x2 v = {sig[2 * k], sig[2 * k + 1]}; if (NT) __builtin_nontemporal_store(v, g_o + k); else g_o[k] = v; }
        }
        if (NT) __builtin_nontemporal_store(dp, dp_out + p0 + lane); else dp_out[p0 + lane] = dp;
        f64x2* g_c = (f64x2*)(C_tang + p0 * (D * D));
#pragma unroll 3
        for (int it = 0; it < CH_CT; ++it) {
            const int q = it * 64 + lane;
            const int pt = q / CH_CT;
            const int k = q - pt * CH_CT;
            const int i = k / CH_VEC;
            const int j0 = (k - i * CH_VEC) * 2;
            const double n_i = Y[pt * ST + i];
            const f64x2 n_j = Y2[pt * (ST / 2) + (j0 >> 1)];
            const f64x2 ab = Y2[pt * (ST / 2) + CH_VEC];
            f64x2 out;
            out.x = c_elas_ij(c, i, j0) - ab.x * (n_i * n_j.x) - ab.y * dev_ij(i, j0);
            out.y = c_elas_ij(c, i, j0 + 1) - ab.x * (n_i * n_j.y) - ab.y * dev_ij(i, j0 + 1);
            if (NT) __builtin_nontemporal_store(out, g_c + q); else g_c[q] = out;
        }
        wave_lds_fence();
    }
}

// pure stream probe with the same 13:43 mix, tile = 64 lanes x (13 + 43) 16-byte chunks
template <bool BLOCKED, bool NT>
__global__ __launch_bounds__(256) void probe(long n_tiles, const f64x2* __restrict__ src, f64x2* __restrict__ dst) {
    constexpr int R = 13, W = 43;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    long t_begin, t_end, t_step;
    if (BLOCKED) { const long per = (n_tiles + nw - 1) / nw; t_begin = gw * per; t_end = min(t_begin + per, n_tiles); t_step = 1; }
    else { t_begin = gw; t_end = n_tiles; t_step = nw; }
    for (long t = t_begin; t < t_end; t += t_step) {
        const f64x2* s = src + t * (R * 64);
        f64x2* d = dst + t * (W * 64);
        f64x2 v[R], acc = {0, 0};
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = s[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < R; ++k) acc += v[k];
#pragma unroll
        for (int k = 0; k < W; ++k) { const f64x2 o = acc + v[k % R]; if (NT) __builtin_nontemporal_store(o, d + k * 64 + lane); else d[k * 64 + lane] = o; }
    }
}

// access-pattern twin of vm6 with no LDS and no arithmetic: 3 input arrays, 3 output arrays, same chunking.
// TILES = consecutive 64-point tiles handled per loop iteration by one wave.
template <int TILES, bool NT>
__global__ __launch_bounds__(256) void probe3(long n, const double* __restrict__ deps, const double* __restrict__ sigma_n,
                                              const double* __restrict__ p, double* __restrict__ C_tang, double* __restrict__ sigma,
                                              double* __restrict__ dp_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n_tiles = n / (64 * TILES);
    const long nw = (long)gridDim.x * 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += nw) {
        const long p0 = tile * 64 * TILES;
        const f64x2* g_e = (const f64x2*)(deps + p0 * 6);
        const f64x2* g_s = (const f64x2*)(sigma_n + p0 * 6);
        f64x2 ve[3 * TILES], vs[3 * TILES];
        double pl[TILES];
#pragma unroll
        for (int k = 0; k < 3 * TILES; ++k) { ve[k] = g_e[k * 64 + lane]; vs[k] = g_s[k * 64 + lane]; }
#pragma unroll
        for (int k = 0; k < TILES; ++k) pl[k] = p[p0 + k * 64 + lane];
        f64x2 acc = {0, 0};
#pragma unroll
        for (int k = 0; k < 3 * TILES; ++k) acc += ve[k] * vs[k];
#pragma unroll
        for (int k = 0; k < TILES; ++k) { const double o = acc.x + pl[k]; if (NT) __builtin_nontemporal_store(o, dp_out + p0 + k * 64 + lane); else dp_out[p0 + k * 64 + lane] = o; }
        f64x2* g_o = (f64x2*)(sigma + p0 * 6);
#pragma unroll
        for (int k = 0; k < 3 * TILES; ++k) { const f64x2 o = acc + ve[k]; if (NT) __builtin_nontemporal_store(o, g_o + k * 64 + lane); else g_o[k * 64 + lane] = o; }
        f64x2* g_c = (f64x2*)(C_tang + p0 * 36);
#pragma unroll
        for (int k = 0; k < 18 * TILES; ++k) { const f64x2 o = acc + vs[k % (3 * TILES)]; if (NT) __builtin_nontemporal_store(o, g_c + k * 64 + lane); else g_c[k * 64 + lane] = o; }
    }
}

// pure read (reduce into one value per lane, written once per wave-run) and pure write streams
template <bool NT>
__global__ __launch_bounds__(256) void pure_write(long n_chunks, f64x2* __restrict__ dst) {
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 * 8) + threadIdx.x; i < n_chunks; i += stride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const f64x2 o = {(double)i, (double)k}; if (i + k * 256 < n_chunks) { if (NT) __builtin_nontemporal_store(o, dst + i + k * 256); else dst[i + k * 256] = o; } }
    }
}
__global__ __launch_bounds__(256) void pure_read(long n_chunks, const f64x2* __restrict__ src, f64x2* __restrict__ sink) {
    const long stride = (long)gridDim.x * 256 * 8;
    f64x2 acc = {0, 0};
    for (long i = ((long)blockIdx.x * 256 * 8) + threadIdx.x; i < n_chunks; i += stride) {
        f64x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k * 256 < n_chunks) ? src[i + k * 256] : f64x2{0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    if (acc.x == 1.2345e301) sink[threadIdx.x] = acc;
}

// bisecting twin: which feature of the vm access pattern costs bandwidth?
template <int TILES, bool USE_P, bool SPLIT_IN, bool SPLIT_OUT>
__global__ __launch_bounds__(256) void probe4(long n, const double* __restrict__ deps, const double* __restrict__ sigma_n,
                                              const double* __restrict__ p, double* __restrict__ C_tang, double* __restrict__ sigma,
                                              double* __restrict__ dp_out, const f64x2* __restrict__ src1, f64x2* __restrict__ dst1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long n_tiles = n / (64 * TILES);
    const long nw = (long)gridDim.x * 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += nw) {
        const long p0 = tile * 64 * TILES;
        f64x2 ve[3 * TILES], vs[3 * TILES];
        double pl[TILES];
        if (SPLIT_IN) {
            const f64x2* g_e = (const f64x2*)(deps + p0 * 6);
            const f64x2* g_s = (const f64x2*)(sigma_n + p0 * 6);
#pragma unroll
            for (int k = 0; k < 3 * TILES; ++k) { ve[k] = g_e[k * 64 + lane]; vs[k] = g_s[k * 64 + lane]; }
        } else {
            const f64x2* g = src1 + tile * (6 * TILES * 64);
#pragma unroll
            for (int k = 0; k < 3 * TILES; ++k) { ve[k] = g[k * 64 + lane]; vs[k] = g[(k + 3 * TILES) * 64 + lane]; }
        }
        if (USE_P) {
#pragma unroll
            for (int k = 0; k < TILES; ++k) pl[k] = p[p0 + k * 64 + lane];
        }
        f64x2 acc = {0, 0};
#pragma unroll
        for (int k = 0; k < 3 * TILES; ++k) acc += ve[k] * vs[k];
        if (USE_P) {
#pragma unroll
            for (int k = 0; k < TILES; ++k) __builtin_nontemporal_store(acc.x + pl[k], dp_out + p0 + k * 64 + lane);
        }
        if (SPLIT_OUT) {
            f64x2* g_o = (f64x2*)(sigma + p0 * 6);
#pragma unroll
            for (int k = 0; k < 3 * TILES; ++k) __builtin_nontemporal_store(acc + ve[k], g_o + k * 64 + lane);
            f64x2* g_c = (f64x2*)(C_tang + p0 * 36);
#pragma unroll
            for (int k = 0; k < 18 * TILES; ++k) __builtin_nontemporal_store(acc + vs[k % (3 * TILES)], g_c + k * 64 + lane);
        } else {
            f64x2* g = dst1 + tile * (21 * TILES * 64);
#pragma unroll
            for (int k = 0; k < 21 * TILES; ++k) __builtin_nontemporal_store(acc + vs[k % (3 * TILES)], g + k * 64 + lane);
        }
    }
}

// write-side bisect: per tile read 6 chunks from src1, write CA chunks to outA + tile*CA KB and CB chunks to outB + tile*CB KB
template <int CA, int CB, bool NT>
__global__ __launch_bounds__(256) void probe5(long n_tiles, const f64x2* __restrict__ src1, f64x2* __restrict__ outA, f64x2* __restrict__ outB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * 4;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += nw) {
        const f64x2* g = src1 + tile * (6 * 64);
        f64x2 v[6], acc = {0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = g[k * 64 + lane];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += v[k];
        f64x2* a = outA + tile * (CA * 64);
        f64x2* b = outB + tile * (CB * 64);
#pragma unroll
        for (int k = 0; k < CA; ++k) { if (NT) __builtin_nontemporal_store(acc + v[k % 6], a + k * 64 + lane); else a[k * 64 + lane] = acc + v[k % 6]; }
#pragma unroll
        for (int k = 0; k < CB; ++k) { if (NT) __builtin_nontemporal_store(acc - v[k % 6], b + k * 64 + lane); else b[k * 64 + lane] = acc - v[k % 6]; }
    }
}

int main(int argc, char** argv) {
    const long n = (argc > 1 ? atol(argv[1]) : 10000000L) / 128 * 128;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5, launches = argc > 3 ? atoi(argv[3]) : 10;
    constexpr int D = 6;
    std::vector<double> h_e(n * D), h_s(n * D), h_p(n);
    srand(1);
    auto rnd = [] { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
    for (long i = 0; i < n * D; ++i) { h_e[i] = rnd() * 5e-3; h_s[i] = rnd() * 170.0; }
    for (long i = 0; i < n; ++i) h_p[i] = fabs(rnd()) * 1e-3;
    double *d_e, *d_s, *d_p, *d_C, *d_sig, *d_dp, *d_C0, *d_src, *d_dst;
    CK(hipMalloc(&d_e, n * D * 8)); CK(hipMalloc(&d_s, n * D * 8)); CK(hipMalloc(&d_p, n * 8));
    CK(hipMalloc(&d_C, n * 36 * 8)); CK(hipMalloc(&d_sig, n * D * 8)); CK(hipMalloc(&d_dp, n * 8));
    CK(hipMalloc(&d_C0, n * 36 * 8));
    const long ptiles = n / 128;
    CK(hipMalloc(&d_src, ptiles * 13 * 1024)); CK(hipMalloc(&d_dst, ptiles * 43 * 1024));
    CK(hipMemcpy(d_e, h_e.data(), n * D * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_s, h_s.data(), n * D * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_p, h_p.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_src, 0, ptiles * 13 * 1024));
    const double E = 70e3, nu = 0.3, H = E * (E / 100) / (E - E / 100), mu = E / 2 / (1 + nu);
    VmConst c{E * nu / (1 + nu) / (1 - 2 * nu), 2 * mu, 3 * mu, 250.0, H, 3 * mu + H, 3 * mu / (3 * mu + H)};
    hipStream_t st; CK(hipStreamCreate(&st));
    const long n_tiles = n / 64;
    const int full_grid = (int)((n_tiles + 3) / 4);
    struct Case { std::string name; std::function<void()> fn; double bytes; std::vector<float> ms; };
    std::vector<Case> cases;
    const double vb = 448.0 * n, pb = (double)ptiles * 56 * 1024;
#define VM(RECIP, LDSMODE, BLOCKED, MINW, NT, GRID, LABEL) cases.push_back({LABEL, [=] { hipLaunchKernelGGL((vm6<RECIP, LDSMODE, BLOCKED, MINW, NT>), dim3(GRID), dim3(256), 0, st, c, n, d_e, d_s, d_p, d_C, d_sig, d_dp); }, vb, {}})
    VM(false, 0, false, 4, true, full_grid, "base div lds7k strided full-grid");
    VM(true, 0, false, 4, true, full_grid, "recip lds7k full-grid");
    VM(true, 0, false, 4, true, 256 * 16, "recip lds7k gridstride bpc16");
    VM(true, 0, true, 4, true, 256 * 16, "recip lds7k blocked bpc16");
    VM(true, 0, true, 4, true, 256 * 5, "recip lds7k blocked bpc5");
    VM(true, 1, false, 4, true, full_grid, "recip lds4k full-grid");
    VM(true, 1, false, 6, true, full_grid, "recip lds4k minw6 full-grid");
    VM(true, 1, true, 6, true, 256 * 6, "recip lds4k minw6 blocked bpc6");
    VM(true, 1, false, 6, true, 256 * 12, "recip lds4k minw6 gridstride bpc12");
    VM(true, 1, false, 8, true, full_grid, "recip lds4k minw8 full-grid");
    VM(true, 1, true, 8, true, 256 * 8, "recip lds4k minw8 blocked bpc8");
    VM(true, 1, false, 6, false, full_grid, "recip lds4k minw6 full-grid noNT");
#define P3(TILES, NT, GRID, LABEL) cases.push_back({LABEL, [=] { hipLaunchKernelGGL((probe3<TILES, NT>), dim3(GRID), dim3(256), 0, st, n, d_e, d_s, d_p, d_C, d_sig, d_dp); }, vb + 1, {}})
    P3(1, true, full_grid, "probe3 t1 full-grid");
    P3(1, true, 4096, "probe3 t1 bpc16");
    P3(1, false, 4096, "probe3 t1 bpc16 noNT");
    P3(2, true, 4096, "probe3 t2 bpc16");
    P3(2, true, full_grid / 2, "probe3 t2 full-grid");
    P3(4, true, 4096, "probe3 t4 bpc16");
    P3(4, true, 2048, "probe3 t4 bpc8");
    const double b_nop = 432.0 * n;
#define P4(TILES, USE_P, SI, SO, GRID, LABEL) cases.push_back({LABEL, [=] { hipLaunchKernelGGL((probe4<TILES, USE_P, SI, SO>), dim3(GRID), dim3(256), 0, st, n, d_e, d_s, d_p, d_C, d_sig, d_dp, (const f64x2*)d_src, (f64x2*)d_dst); }, (USE_P ? vb : b_nop) + 3, {}})
    P4(1, true, true, true, 4096, "probe4 p split-in split-out (=probe3)");
    P4(1, false, true, true, 4096, "probe4 nop split-in split-out");
    P4(1, false, false, true, 4096, "probe4 nop one-in split-out");
    P4(1, false, true, false, 4096, "probe4 nop split-in one-out");
    P4(1, false, false, false, 4096, "probe4 nop one-in one-out");
    P4(2, false, false, false, 4096, "probe4 nop one-in one-out t2");
    P4(1, true, false, false, 4096, "probe4 p one-in one-out");
    const long nt64 = n / 64;
    const double b5 = nt64 * 27.0 * 1024 + 5;
    double *bufA, *bufB;  // generously sized separate allocations for the write-side bisect
    CK(hipMalloc(&bufA, nt64 * 22 * 1024)); CK(hipMalloc(&bufB, nt64 * 22 * 1024));
    f64x2* sameB = (f64x2*)bufA + nt64 * 3 * 64;  // second region inside allocation A (for the 3+18 split)
#define P5(CA, CB, NT, A, B, LABEL) cases.push_back({LABEL, [=] { hipLaunchKernelGGL((probe5<CA, CB, NT>), dim3(4096), dim3(256), 0, st, nt64, (const f64x2*)d_src, (f64x2*)(A), (f64x2*)(B)); }, b5, {}})
    P5(21, 0, true, bufA, bufA, "probe5 21+0 -> A");
    P5(21, 0, true, bufB, bufB, "probe5 21+0 -> B");
    P5(3, 18, true, bufA, bufB, "probe5 3+18 -> A,B");
    P5(3, 18, true, bufA, sameB, "probe5 3+18 -> A,A+off");
    P5(18, 3, true, bufA, bufB, "probe5 18+3 -> A,B");
    P5(3, 18, false, bufA, bufB, "probe5 3+18 -> A,B noNT");
    P5(4, 17, true, bufA, bufB, "probe5 4+17 -> A,B");
    P5(2, 19, true, bufA, bufB, "probe5 2+19 -> A,B");
    P5(1, 20, true, bufA, bufB, "probe5 1+20 -> A,B");
    P5(8, 13, true, bufA, bufB, "probe5 8+13 -> A,B");
    P5(16, 5, true, bufA, bufB, "probe5 16+5 -> A,B");
    P5(10, 11, true, bufA, bufB, "probe5 10+11 -> A,B");
    const long wchunks = (long)(n * 36 / 2);  // C_tang-sized buffer, 2.88 GB
    cases.push_back({"pure write NT 2.88GB full", [=] { hipLaunchKernelGGL((pure_write<true>), dim3((int)((wchunks + 2047) / 2048)), dim3(256), 0, st, wchunks, (f64x2*)d_C); }, wchunks * 16.0, {}});
    cases.push_back({"pure write    2.88GB full", [=] { hipLaunchKernelGGL((pure_write<false>), dim3((int)((wchunks + 2047) / 2048)), dim3(256), 0, st, wchunks, (f64x2*)d_C); }, wchunks * 16.0, {}});
    cases.push_back({"pure write NT 2.88GB bpc16", [=] { hipLaunchKernelGGL((pure_write<true>), dim3(4096), dim3(256), 0, st, wchunks, (f64x2*)d_C); }, wchunks * 16.0, {}});
    cases.push_back({"pure read 2.88GB full", [=] { hipLaunchKernelGGL(pure_read, dim3((int)((wchunks + 2047) / 2048)), dim3(256), 0, st, wchunks, (const f64x2*)d_C0, (f64x2*)d_dst); }, wchunks * 16.0, {}});
    cases.push_back({"pure read 2.88GB bpc16", [=] { hipLaunchKernelGGL(pure_read, dim3(4096), dim3(256), 0, st, wchunks, (const f64x2*)d_C0, (f64x2*)d_dst); }, wchunks * 16.0, {}});
    cases.push_back({"probe gridstride bpc16", [=] { hipLaunchKernelGGL((probe<false, true>), dim3(4096), dim3(256), 0, st, ptiles, (const f64x2*)d_src, (f64x2*)d_dst); }, pb + 2, {}});
    cases.push_back({"probe blocked bpc16", [=] { hipLaunchKernelGGL((probe<true, true>), dim3(4096), dim3(256), 0, st, ptiles, (const f64x2*)d_src, (f64x2*)d_dst); }, pb + 2, {}});
    cases.push_back({"probe blocked bpc8", [=] { hipLaunchKernelGGL((probe<true, true>), dim3(2048), dim3(256), 0, st, ptiles, (const f64x2*)d_src, (f64x2*)d_dst); }, pb + 2, {}});
    cases.push_back({"probe full-grid", [=] { hipLaunchKernelGGL((probe<false, true>), dim3((int)((ptiles + 3) / 4)), dim3(256), 0, st, ptiles, (const f64x2*)d_src, (f64x2*)d_dst); }, pb + 2, {}});

    // correctness of every vm variant against the first
    cases[0].fn(); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(d_C0, d_C, n * 36 * 8, hipMemcpyDeviceToDevice));
    std::vector<double> ref(4096 * 36), got(4096 * 36);
    CK(hipMemcpy(ref.data(), d_C0 + (n / 2) * 36, ref.size() * 8, hipMemcpyDeviceToHost));
    for (auto& cs : cases) {
        if (cs.bytes != vb) continue;
        CK(hipMemset(d_C, 0, n * 36 * 8));
        cs.fn(); CK(hipStreamSynchronize(st)); CK(hipGetLastError());
        CK(hipMemcpy(got.data(), d_C + (n / 2) * 36, got.size() * 8, hipMemcpyDeviceToHost));
        double err = 0, scale = 0;
        for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, fabs(got[i] - ref[i])); scale = std::max(scale, fabs(ref[i])); }
        // also the very last tile
        std::vector<double> tail_r(64 * 36), tail_g(64 * 36);
        CK(hipMemcpy(tail_r.data(), d_C0 + (n - 64) * 36, tail_r.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(tail_g.data(), d_C + (n - 64) * 36, tail_g.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tail_r.size(); ++i) err = std::max(err, fabs(tail_g[i] - tail_r[i]));
        if (!(err <= 1e-13 * scale)) printf("MISMATCH %-40s rel err %.3e\n", cs.name.c_str(), err / scale);
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < rounds; ++r)
        for (auto& cs : cases) {
            CK(hipEventRecord(a, st));
            for (int l = 0; l < launches; ++l) cs.fn();
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            cs.ms.push_back(ms / launches);
        }
    printf("n=%ld bytes/launch %.3f GB\n", n, vb / 1e9);
    for (auto& cs : cases) {
        std::sort(cs.ms.begin(), cs.ms.end());
        const float med = cs.ms[cs.ms.size() / 2], best = cs.ms[0];
        printf("%-42s median %.4f ms %8.1f GB/s   best %8.1f GB/s\n", cs.name.c_str(), med, cs.bytes / med / 1e6, cs.bytes / best / 1e6);
    }
    return 0;
}
