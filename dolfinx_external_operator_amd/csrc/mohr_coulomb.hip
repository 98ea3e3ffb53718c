// mohr_coulomb.hip — Mohr-Coulomb (Abbo-Sloan smoothed) return mapping + forward-mode-through-Newton
// tangent, one lane per quadrature point. The per-point algorithm is csrc/mc_core.h (closed-form
// restatement of the reference's nested jacfwd, doc/demo/demo_plasticity_mohr_coulomb.py:282-555).
//
// Roofline: NOT HBM. Algorithmic traffic is 224 B/point (read 4+4, write 16+4 doubles; +28 B with the
// four diagnostics) against ~3-8 kflop of fp64 per Newton iteration and 1 (elastic) / 3-10 (plastic)
// iterations: FP64-VALU- and divergence-bound. HBM GB/s is still reported by the bench.
//
// Memory side: deps / sigma_n are 32 B per point (two 16-byte loads per lane, both halves of every
// 128-byte line are consumed by the same wave). C_tang (128 B per point) and sigma go through a
// wave-private LDS slice so every global store instruction writes 1 KiB of consecutive bytes; the
// diagnostics are naturally lane-linear.
//
// Divergence: like vmap(while_loop) (:564-571) a wave iterates until its slowest lane has converged;
// converged lanes are masked off, so per-point results equal the scalar algorithm.
#include "dxo_common.h"
#include "mc_core.h"

namespace {

constexpr int MC_PENDING = 512;  // per-wave buffer of plastic point indices (mc_classify)
#ifndef DXO_MC_SIGMA_ALL
#define DXO_MC_SIGMA_ALL 1    // mc_classify: whole-line sigma stores for every point (mc_newton overwrites the plastic ones): 1.30 vs 1.35 ms
#endif
#ifndef DXO_MC_LDS_STATE
#define DXO_MC_LDS_STATE 1    // mc_newton: inputs and tangent iterate of a lane's point parked in LDS between their uses
#endif
constexpr int MC_BATCH = 256;    // list entries a wave reserves per cursor atomic (mc_newton); 128 and 64 were not faster
constexpr int MC_ROW = 18;  // LDS doubles per point: 16 C_tang + 2 pad (144 B stride: conflict-free b128 writes)

template <bool NT>
__device__ __forceinline__ void st16(dxo_f64x2* p, dxo_f64x2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__global__ __launch_bounds__(DXO_BLOCK) void mc_point(mc::Const k, int64_t n, const double* __restrict__ deps,
                                                      const double* __restrict__ sigma_n, double* __restrict__ C_tang,
                                                      double* __restrict__ sigma, int32_t* __restrict__ niter,
                                                      double* __restrict__ yielding, double* __restrict__ norm_res,
                                                      double* __restrict__ dlambda) {
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * DXO_WAVE * MC_ROW];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* X = lds + wave * (DXO_WAVE * MC_ROW);
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t p0 = tile * DXO_WAVE;
        const int npts = (n - p0 < DXO_WAVE) ? (int)(n - p0) : DXO_WAVE;
        const bool live = lane < npts;
        const int64_t i = p0 + (live ? lane : 0);
        const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + i * 4);
        const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + i * 4);
        const dxo_f64x2 e01 = ge[0], e23 = ge[1], s01 = gs[0], s23 = gs[1];
        const double e[4] = {e01.x, e01.y, e23.x, e23.y};
        const double s[4] = {s01.x, s01.y, s23.x, s23.y};
        mc::Result R;
        mc::return_map(k, e, s, R);
        // ---- outputs through the wave's LDS slice: point-per-lane rows -> lane-linear 16-byte stores
#pragma unroll
        for (int c = 0; c < 8; ++c) X2[(lane * MC_ROW) / 2 + c] = dxo_f64x2{R.C_tang[2 * c], R.C_tang[2 * c + 1]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + p0 * 16);
        const int nct = npts * 8;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = it * DXO_WAVE + lane;
            const int pt = q >> 3, c = q & 7;
            const dxo_f64x2 v = X2[(pt * MC_ROW) / 2 + c];
            if (q < nct) st16<true>(gc + q, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        X2[lane * 2] = dxo_f64x2{R.sigma[0], R.sigma[1]};
        X2[lane * 2 + 1] = dxo_f64x2{R.sigma[2], R.sigma[3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + p0 * 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * DXO_WAVE + lane;
            if (q < npts * 2) st16<true>(gg + q, X2[q]);
        }
        if (live) {
            if (niter) niter[i] = R.niter;
            if (yielding) yielding[i] = R.yielding;
            if (norm_res) norm_res[i] = R.norm_res;
            if (dlambda) dlambda[i] = R.dlambda;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ------------------------------------------------------------------ variant 1: classify + compacted Newton
// The simple kernel above makes every wave run as many Newton passes as its slowest lane while elastic
// lanes (1 "iteration") idle. On a mixed workload (BASELINE config 4: ~30 % plastic, 2-5 iterations) that
// is < 20 % lane utilisation. Variant 1 splits the work:
//   mc_classify  lane = point, HBM-bound: trial stress, f(trial); elastic points are finished here
//                (C_elas written in output order straight from constants — no LDS transposition needed —
//                sigma through LDS); plastic points are appended to a compact index list
//                (one wave-aggregated atomic per wave).
//   mc_newton    FP64-bound: waves pull plastic points from the list through a shared cursor. A lane keeps
//                its point until cond_fun fails, writes the point's outputs and is REFILLED with the next
//                list entry while its neighbours keep iterating: every Newton pass runs on (nearly) full
//                waves regardless of how iteration counts are distributed.
// Scratch (ctx-owned): header {uint32 n_plastic, uint32 cursor} + int32 list[n].
struct McScratchHeader {
    unsigned int n_plastic;
    unsigned int cursor;
    unsigned int pad[62];
};

#ifndef DXO_MC_CLASSIFY_MINW
#define DXO_MC_CLASSIFY_MINW 1
#endif
__global__ __launch_bounds__(DXO_BLOCK, DXO_MC_CLASSIFY_MINW) void mc_classify(mc::Const k, int64_t n, const double* __restrict__ deps,
                                                         const double* __restrict__ sigma_n, double* __restrict__ C_tang,
                                                         double* __restrict__ sigma, int32_t* __restrict__ niter,
                                                         double* __restrict__ yielding, double* __restrict__ norm_res,
                                                         double* __restrict__ dlambda, McScratchHeader* __restrict__ hdr,
                                                         int32_t* __restrict__ list) {
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ __attribute__((aligned(16))) double lds[WAVES * DXO_WAVE * 4];
    __shared__ int32_t pending[WAVES * MC_PENDING];   // plastic indices waiting for one batched list append
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(lds + wave * (DXO_WAVE * 4));
    int32_t* pend = pending + wave * MC_PENDING;
    int n_pend = 0;                                    // wave-uniform
    auto flush = [&]() {
        // ONE global atomic per ~MC_PENDING plastic points (a per-tile atomic on one address serialises:
        // 156 k same-address atomics cost 1.8 ms at 10^7 points), and the list append is lane-linear.
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&hdr->n_plastic, (unsigned int)n_pend);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int j = lane; j < n_pend; j += DXO_WAVE) list[base + j] = pend[j];
        n_pend = 0;
    };
    const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t p0 = tile * DXO_WAVE;
        const int npts = (n - p0 < DXO_WAVE) ? (int)(n - p0) : DXO_WAVE;
        const bool live = lane < npts;
        const int64_t i = p0 + (live ? lane : 0);
        const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + i * 4);
        const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + i * 4);
        const dxo_f64x2 e01 = ge[0], e23 = ge[1], s01 = gs[0], s23 = gs[1];
        const double e[4] = {e01.x, e01.y, e23.x, e23.y};
        const double sn[4] = {s01.x, s01.y, s23.x, s23.y};
        double Ce[4], trial[4];
        mc::C_times(k, e, Ce);
#pragma unroll
        for (int c = 0; c < 4; ++c) trial[c] = sn[c] + Ce[c];
        const double yld = mc::f_value(k, trial);                      // :422
        const bool elastic = yld <= 0.0;                               // NaN -> plastic branch, as lax.cond does
        mc::Result R;
        mc::elastic_point(k, sn, Ce, trial, R);
        const unsigned long long el_mask = __ballot(live && elastic);
        const unsigned long long zero_mask = __ballot(live && elastic && R.niter == 0);
        const unsigned long long pl_mask = __ballot(live && !elastic);
        // plastic points -> the wave's pending buffer (ballot-compacted), flushed in batches
        if (pl_mask) {
            if (n_pend + DXO_WAVE > MC_PENDING) flush();
            if (live && !elastic) pend[n_pend + __popcll(pl_mask & ((1ull << lane) - 1ull))] = (int32_t)i;
            n_pend += __popcll(pl_mask);
        }
        if (live) {
            if (yielding) yielding[i] = yld;
#if DXO_MC_SIGMA_ALL
            // whole-line stores for every point: the plastic points' entries are placeholders that mc_newton overwrites
            if (niter) niter[i] = R.niter;
            if (norm_res) norm_res[i] = R.norm_res;
            if (dlambda) dlambda[i] = 0.0;
#else
            if (elastic) {
                if (niter) niter[i] = R.niter;
                if (norm_res) norm_res[i] = R.norm_res;
                if (dlambda) dlambda[i] = 0.0;
            }
#endif
        }
        // sigma of elastic points: point-per-lane rows -> lane-linear stores, masked by the owner's branch
        X2[lane * 2] = dxo_f64x2{R.sigma[0], R.sigma[1]};
        X2[lane * 2 + 1] = dxo_f64x2{R.sigma[2], R.sigma[3]};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + p0 * 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int q = it * DXO_WAVE + lane;
#if DXO_MC_SIGMA_ALL
            if (q < npts * 2) st16<true>(gg + q, X2[q]);   // plastic points get their trial stress here and the returned one from mc_newton
#else
            if ((el_mask >> (q >> 1)) & 1ull) st16<true>(gg + q, X2[q]);
#endif
        }
        // C_tang of elastic points: constants in output order (chunk q = point q/8, entries 2(q%8), 2(q%8)+1)
        dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + p0 * 16);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int q = it * DXO_WAVE + lane;
            const int pt = q >> 3, c = q & 7;
            const int row = c >> 1, col = (c & 1) * 2;
            dxo_f64x2 v;
            v.x = ((row < 3 && col < 3) ? k.lmbda : 0.0) + (row == col ? k.mu2 : 0.0);
            v.y = ((row < 3 && col + 1 < 3) ? k.lmbda : 0.0) + (row == col + 1 ? k.mu2 : 0.0);
            if ((zero_mask >> pt) & 1ull) v = dxo_f64x2{0.0, 0.0};
            if ((el_mask >> pt) & 1ull) st16<true>(gc + q, v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (n_pend) flush();
}

// The inputs and the tangent iterate of a lane's point in the wave's LDS slice: slot-major, lane-minor (a slot of all 64
// lanes is 512 contiguous bytes: conflict-free ds_read/write_b64), slots 0-7 = deps, sn, slots 8 + 5 m + i = Y[i][m].
struct LaneLds {
    double* slots;   // wave slice + lane
    __device__ __forceinline__ void set_inputs(const double* d, const double* s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { slots[i * DXO_WAVE] = d[i]; slots[(4 + i) * DXO_WAVE] = s[i]; }
    }
    __device__ __forceinline__ void get_inputs(double* d, double* s) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { d[i] = slots[i * DXO_WAVE]; s[i] = slots[(4 + i) * DXO_WAVE]; }
    }
    __device__ __forceinline__ void get_col(int m, double* v5) const {
#pragma unroll
        for (int i = 0; i < 5; ++i) v5[i] = slots[(8 + 5 * m + i) * DXO_WAVE];
    }
    __device__ __forceinline__ void set_col(int m, const double* v5) {
#pragma unroll
        for (int i = 0; i < 5; ++i) slots[(8 + 5 * m + i) * DXO_WAVE] = v5[i];
    }
};
constexpr int MC_LANE_SLOTS = 28;

template <int MINW, bool SAME>
__global__ __launch_bounds__(DXO_BLOCK, MINW) void mc_newton(mc::Const k, const double* __restrict__ deps,
                                                       const double* __restrict__ sigma_n, double* __restrict__ C_tang,
                                                       double* __restrict__ sigma, int32_t* __restrict__ niter,
                                                       double* __restrict__ norm_res, double* __restrict__ dlambda,
                                                       McScratchHeader* __restrict__ hdr, const int32_t* __restrict__ list) {
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const unsigned int total = hdr->n_plastic;   // written by mc_classify, earlier on the same stream
#if DXO_MC_LDS_STATE
    __shared__ double lane_state[(DXO_BLOCK / DXO_WAVE) * MC_LANE_SLOTS * DXO_WAVE];
    mc::LaneT<LaneLds> L;
    L.st.slots = lane_state + (threadIdx.x >> 6) * (MC_LANE_SLOTS * DXO_WAVE) + lane;
#else
    mc::Lane L;
#endif
    bool active = false, exhausted = false;
    int64_t idx = 0;
    unsigned int lo = 0, hi = 0;   // the wave's reserved slice of the list (wave-uniform)
    for (;;) {
        // ---- refill idle lanes from the wave's reserved slice; reserve MC_BATCH more entries with ONE atomic
        // when it runs dry (a cursor atomic per pass per wave serialises on one address like n_plastic did)
        const unsigned long long idle = __ballot(!active);
        if (idle && lo == hi && !exhausted) {
            unsigned int start = 0;
            if (lane == 0) start = atomicAdd(&hdr->cursor, (unsigned int)MC_BATCH);
            start = __builtin_amdgcn_readfirstlane(start);
            if (start >= total) exhausted = true;
            else { lo = start; hi = (start + MC_BATCH < total) ? start + MC_BATCH : total; }
        }
        if (idle && lo < hi) {
            const unsigned int mine = lo + (unsigned int)__popcll(idle & ((1ull << lane) - 1ull));
            const unsigned int take = ((unsigned int)__popcll(idle) < hi - lo) ? (unsigned int)__popcll(idle) : hi - lo;
            lo += take;
            if (!active && mine < hi) {
                idx = list[mine];
                const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + idx * 4);
                const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + idx * 4);
                const dxo_f64x2 e01 = ge[0], e23 = ge[1], s01 = gs[0], s23 = gs[1];
                const double e[4] = {e01.x, e01.y, e23.x, e23.y};
                const double sn[4] = {s01.x, s01.y, s23.x, s23.y};
                mc::lane_init(L, e, sn);
                active = true;
            }
        }
        if (!__ballot(active)) break;
        // ---- one Newton pass on every lane that holds a point
        if (active) {
            if (mc::lane_pass<SAME>(k, L)) {
                dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + idx * 16);
                double Yc[4][5];
#pragma unroll
                for (int m = 0; m < 4; ++m) L.st.get_col(m, Yc[m]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gc[2 * r] = dxo_f64x2{Yc[0][r], Yc[1][r]};
                    gc[2 * r + 1] = dxo_f64x2{Yc[2][r], Yc[3][r]};
                }
                dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + idx * 4);
                gg[0] = dxo_f64x2{L.sig[0], L.sig[1]};
                gg[1] = dxo_f64x2{L.sig[2], L.sig[3]};
                if (niter) niter[idx] = L.niter;
                if (norm_res) norm_res[idx] = L.norm;
                if (dlambda) dlambda[idx] = L.dl;
                active = false;
            }
        }
    }
}

// ------------------------------------------------------------------ variant 2: classification and Newton in ONE persistent kernel
// mc_classify is HBM-bound with an idle vector pipe, mc_newton fp64-bound with an idle memory system, and they run one
// after the other. Here every persistent wave does both: it classifies tiles of 64 points (elastic points are finished and
// stored at once, plastic points go on the wave's own LDS stack, index and inputs) whenever the stack cannot fill its idle
// Newton lanes, refills idle lanes from it (LDS reads only), and runs one Newton pass on the lanes that hold a point. A tile's 16 KB of loads and
// stores are in flight while the SIMD's other wave iterates; there is no index list, no cursor atomic, no scratch.
// Tiles are dealt round-robin over all waves (wave w gets tiles w, w + W, w + 2W, ...): whatever the spatial pattern of the
// plastic zone, every wave samples the whole batch evenly.
// Elastic outputs use masked stores (a plastic point's rows are written once, by its Newton lane: two stores of one wave
// to one address from different lanes have no guaranteed order). The per-point arithmetic and its order are those of the
// other variants: outputs are bit-identical.
// Tried on this kernel and dropped (10^7 points, 31 % plastic; this form 1.006 ms, variant 1 1.09 ms on the same board):
//   * the next tile's inputs fetched global -> LDS (global_load_lds) during the Newton pass, counted vmcnt waits: 1.014 ms
//     (the register file is full: the spilled registers' scratch traffic waits on the same counter);
//   * two or four tiles classified per visit with their loads issued first: 1.15 / 1.34 ms (spills);
//   * the pass split into evaluation and step, finished lanes refilled between the two so that the step always runs on a
//     full wave of unfinished points (a point with k steps holds its lane for k instead of k + 1 iterations): 1.045 ms at
//     31 % plastic, 2.33 against 2.31 ms with every point plastic (22 spilled registers). Again after the Lode angle's
//     library calls were gone (no spills): 0.915 against 0.943 ms at 31 %, 1.85 against 1.96 ms all plastic, 0.665 against
//     0.633 ms at 5 % — and the two inlined copies of the pass contract their FMAs differently, so the outputs were no
//     longer bit-identical to the other variants. Not taken.
//   * (after the pass had lost its scratch traffic) the global_load_lds prefetch again: 0.806 against 0.806 ms; and an
//     iteration ordered so that every vmcnt wait sits directly after a pass and all stores are issued in one burst before
//     the next (loads and stores retire through one in-order counter on gfx950, so a wait for a load also waits for
//     the stores before it): 0.825 against 0.806 ms. Neither the load latency nor the store acknowledgements are what the
//     mix waits for.
#ifndef DXO_MC_PROF
#define DXO_MC_PROF 0   // 1: instrumented build for scripts/exp/archive/mc_phase_profile.py only (it corrupts dlambda)
#endif
constexpr int MC_QCAP = 128;   // plastic point indices waiting per wave: at most 63 + 64 from one more tile
constexpr int MC_STASH = 64;   // of which the newest keep their inputs in LDS

template <int MINW, bool SAME>
__global__ __launch_bounds__(DXO_BLOCK, MINW) void mc_fused(mc::Const k, int64_t n, const double* __restrict__ deps,
                                                      const double* __restrict__ sigma_n, double* __restrict__ C_tang,
                                                      double* __restrict__ sigma, int32_t* __restrict__ niter,
                                                      double* __restrict__ yielding, double* __restrict__ norm_res,
                                                      double* __restrict__ dlambda) {
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    __shared__ double lane_state[WAVES * MC_LANE_SLOTS * DXO_WAVE];
    __shared__ int32_t queue[WAVES * MC_QCAP];
    // the inputs (deps, sigma_n: 64 bytes) of the waiting points, written by the classifying lane that has them in registers, so
    // that a refill is four LDS reads instead of a gather from L2 / HBM: slot = stack position mod 64, chunk-major (conflict-free)
    __shared__ __attribute__((aligned(16))) dxo_f64x2 stash[WAVES * 4 * MC_STASH];
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    int32_t* q = queue + wave * MC_QCAP;
    dxo_f64x2* sth = stash + wave * (4 * MC_STASH);
    mc::LaneT<LaneLds> L;
    L.st.slots = lane_state + wave * (MC_LANE_SLOTS * DXO_WAVE) + lane;
    bool active = false;
    int32_t idx = 0;               // n <= 2^30 per launch (mc_launch splits larger batches)
    // The waiting points are a STACK (positions 0 .. q_count - 1; the newest leave first — which lane iterates a point, and when,
    // does not change its result). Stack position p keeps its inputs in stash slot p mod 64, so the newest 64 entries always
    // have theirs; an entry below `stash_lo` has had its slot overwritten by position p + 64 and is re-read from global memory.
    int q_count = 0, stash_lo = 0;   // wave-uniform
    const int n_tiles = (int)((n + DXO_WAVE - 1) / DXO_WAVE);
    const int stride = (int)gridDim.x * WAVES;
    int tile = __builtin_amdgcn_readfirstlane((int)blockIdx.x * WAVES + wave);
    auto store_point = [&]() {
        dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + (int64_t)idx * 16);
        double Yc[4][5];
#pragma unroll
        for (int m = 0; m < 4; ++m) L.st.get_col(m, Yc[m]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gc[2 * r] = dxo_f64x2{Yc[0][r], Yc[1][r]};
            gc[2 * r + 1] = dxo_f64x2{Yc[2][r], Yc[3][r]};
        }
        dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + (int64_t)idx * 4);
        gg[0] = dxo_f64x2{L.sig[0], L.sig[1]};
        gg[1] = dxo_f64x2{L.sig[2], L.sig[3]};
        if (niter) niter[idx] = L.niter;
        if (norm_res) norm_res[idx] = L.norm;
        if (dlambda) dlambda[idx] = L.dl;
    };
#if DXO_MC_PROF
    // EXPERIMENT ONLY (scripts/exp/archive/mc_phase_profile.py): cycles per wave in classification / refill / Newton pass, number of passes
    long long prof[4] = {0, 0, 0, 0};
    const long long prof_t0 = __builtin_readcyclecounter();
    long long prof_t = prof_t0;
#define DXO_PROF_MARK(slot) { const long long t_ = __builtin_readcyclecounter(); prof[slot] += t_ - prof_t; prof_t = t_; }
#else
#define DXO_PROF_MARK(slot)
#endif
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        // ---- classify tiles until the queue can fill the idle lanes (or the tiles run out)
        while (q_count < n_idle && tile < n_tiles) {
            const int64_t p0 = (int64_t)tile * DXO_WAVE;
            tile += stride;
            const int npts = (n - p0 < DXO_WAVE) ? (int)(n - p0) : DXO_WAVE;
            const bool live = lane < npts;
            const int64_t i = p0 + (live ? lane : 0);
            const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + i * 4);
            const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + i * 4);
            const dxo_f64x2 e01 = ge[0], e23 = ge[1], s01 = gs[0], s23 = gs[1];
            const double e[4] = {e01.x, e01.y, e23.x, e23.y};
            const double sn[4] = {s01.x, s01.y, s23.x, s23.y};
            double Ce[4], trial[4];
            mc::C_times(k, e, Ce);
#pragma unroll
            for (int c = 0; c < 4; ++c) trial[c] = sn[c] + Ce[c];
            const double yld = mc::f_value(k, trial, SAME ? 1 : 0);         // :422
            const bool elastic = yld <= 0.0;                               // NaN -> plastic branch, as lax.cond does
            mc::Result R;
            mc::elastic_point(k, sn, Ce, trial, R);
            const unsigned long long el_mask = __ballot(live && elastic);
            const unsigned long long zero_mask = __ballot(live && elastic && R.niter == 0);
            const unsigned long long pl_mask = __ballot(live && !elastic);
            if (live && !elastic) {
                const int pos = q_count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pl_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pl_mask, 0u));
                q[pos] = (int32_t)i;
                dxo_f64x2* st = sth + (pos & (MC_STASH - 1));
                st[0] = e01; st[MC_STASH] = e23; st[2 * MC_STASH] = s01; st[3 * MC_STASH] = s23;
            }
            q_count += __popcll(pl_mask);
            if (q_count - MC_STASH > stash_lo) stash_lo = q_count - MC_STASH;
            if (live) {
                if (yielding) yielding[i] = yld;
                if (elastic) {
                    if (niter) niter[i] = R.niter;
                    if (norm_res) norm_res[i] = R.norm_res;
                    if (dlambda) dlambda[i] = 0.0;
                    // the lane's own 32-byte row (the neighbours' rows complete the lines in L2)
                    dxo_f64x2* gg = reinterpret_cast<dxo_f64x2*>(sigma + i * 4);
                    st16<true>(gg, dxo_f64x2{R.sigma[0], R.sigma[1]});
                    st16<true>(gg + 1, dxo_f64x2{R.sigma[2], R.sigma[3]});
                }
            }
            // C_tang of elastic points: constants in output order (chunk c = point c/8, entries 2(c%8), 2(c%8)+1)
            dxo_f64x2* gc = reinterpret_cast<dxo_f64x2*>(C_tang + p0 * 16);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int c = it * DXO_WAVE + lane;
                const int cc = lane & 7, sub = lane >> 3;            // chunk c belongs to point 8 it + sub
                const int row = cc >> 1, col = (cc & 1) * 2;
                dxo_f64x2 v;
                v.x = ((row < 3 && col < 3) ? k.lmbda : 0.0) + (row == col ? k.mu2 : 0.0);
                v.y = ((row < 3 && col + 1 < 3) ? k.lmbda : 0.0) + (row == col + 1 ? k.mu2 : 0.0);
                // the eight points of this iteration: one byte of each mask (a scalar shift), tested with a 32-bit vector shift
                const unsigned zero8 = (unsigned)(zero_mask >> (8 * it)) & 0xffu, el8 = (unsigned)(el_mask >> (8 * it)) & 0xffu;
                if ((zero8 >> sub) & 1u) v = dxo_f64x2{0.0, 0.0};
                if ((el8 >> sub) & 1u) st16<true>(gc + c, v);
            }
        }
        DXO_PROF_MARK(0)
        // ---- refill idle lanes from the queue
        if (n_idle && q_count) {
            const int rank = __popcll(idle & ((1ull << lane) - 1ull));
            const int take = n_idle < q_count ? n_idle : q_count;
            if (!active && rank < take) {
                const int pos = q_count - 1 - rank;   // from the top of the stack
                idx = q[pos];
                dxo_f64x2 e01, e23, s01, s23;
                if (pos >= stash_lo) {
                    const dxo_f64x2* st = sth + (pos & (MC_STASH - 1));
                    e01 = st[0]; e23 = st[MC_STASH]; s01 = st[2 * MC_STASH]; s23 = st[3 * MC_STASH];
                } else {
                    const dxo_f64x2* ge = reinterpret_cast<const dxo_f64x2*>(deps + (int64_t)idx * 4);
                    const dxo_f64x2* gs = reinterpret_cast<const dxo_f64x2*>(sigma_n + (int64_t)idx * 4);
                    e01 = ge[0]; e23 = ge[1]; s01 = gs[0]; s23 = gs[1];
                }
                const double e[4] = {e01.x, e01.y, e23.x, e23.y};
                const double sn[4] = {s01.x, s01.y, s23.x, s23.y};
                mc::lane_init(L, e, sn);
                active = true;
            }
            q_count -= take;
            if (q_count < stash_lo) stash_lo = q_count;   // everything below the old mark is still stale, everything pushed from here on is fresh
        }
        DXO_PROF_MARK(1)
        if (!__ballot(active)) break;   // nothing waiting either: the queue would have refilled, the tiles would have been classified
        // ---- one Newton pass on every lane that holds a point
        if (active) {
            if (mc::lane_pass<SAME>(k, L)) {
                store_point();
                active = false;
            }
        }
        DXO_PROF_MARK(2)
#if DXO_MC_PROF
        prof[3] += 1;
#endif
    }
#if DXO_MC_PROF
    if (lane == 0 && dlambda) {   // overwrites the head of dlambda with the wave's counters: the results of such a build are NOT the operator's
        const int64_t w = (int64_t)blockIdx.x * WAVES + wave;
        __builtin_amdgcn_s_waitcnt(0);
        for (int c = 0; c < 4; ++c) dlambda[w * 5 + c] = (double)prof[c];
        dlambda[w * 5 + 4] = (double)(__builtin_readcyclecounter() - prof_t0);
    }
#endif
}

struct McLaunch {
    mc::Const k;
    bool d_niter, d_yield, d_res, d_dl;
};

void* mc_scratch(dxo_ctx* ctx, hipStream_t s, size_t bytes) { return dxo_scratch(ctx, s, bytes); }

int mc_launch(dxo_ctx* ctx, const McLaunch& L, int64_t n, const double* deps, const double* sigma_n, double* C_tang,
              double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda, hipStream_t s) {
    if (n == 0) return DXO_OK;
    if (ctx->mc_variant == 0) {
        const int64_t n_tiles = (n + DXO_WAVE - 1) / DXO_WAVE;
        const int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
        hipLaunchKernelGGL(mc_point, dim3(grid), dim3(DXO_BLOCK), 0, s, L.k, n, deps, sigma_n, C_tang, sigma, niter,
                           yielding, norm_res, dlambda);
        return DXO_OK;
    }
    if (ctx->mc_variant == 2) {
        // one persistent kernel, two workgroups per CU (all resident: 248 registers and 66 KB of LDS each)
        const int64_t max_idx = (int64_t)1 << 30;   // queue entries are int32
        for (int64_t off = 0; off < n; off += max_idx) {
            const int64_t m = (n - off < max_idx) ? (n - off) : max_idx;
            int64_t blocks = (int64_t)ctx->compute_units * 2;
            const int64_t enough = (m + DXO_BLOCK - 1) / DXO_BLOCK;
            if (blocks > enough) blocks = enough;
#define DXO_MC_FUSED(S_)                                                                                                       \
    hipLaunchKernelGGL((mc_fused<2, S_>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, L.k, m, deps + off * 4, sigma_n + off * 4,   \
                       C_tang + off * 16, sigma + off * 4, niter ? niter + off : nullptr, yielding ? yielding + off : nullptr,  \
                       norm_res ? norm_res + off : nullptr, dlambda ? dlambda + off : nullptr)
            if (L.k.same_angle != 0) DXO_MC_FUSED(true); else DXO_MC_FUSED(false);
#undef DXO_MC_FUSED
        }
        return DXO_OK;
    }
    // int32 list entries: split gigantic batches
    int64_t max_part = ctx->mc_part_points;
    if (max_part < DXO_WAVE) max_part = DXO_WAVE;
    if (max_part > ((int64_t)1 << 30)) max_part = (int64_t)1 << 30;
    max_part = max_part / DXO_WAVE * DXO_WAVE;      // parts start on 16-byte aligned rows
    for (int64_t off = 0; off < n; off += max_part) {
        const int64_t m = (n - off < max_part) ? (n - off) : max_part;
        void* scratch = mc_scratch(ctx, s, sizeof(McScratchHeader) + (size_t)m * sizeof(int32_t));
        if (!scratch) return dxo_hip_fail(ctx, hipErrorOutOfMemory, "dxo_mohr_coulomb: scratch allocation");
        McScratchHeader* hdr = static_cast<McScratchHeader*>(scratch);
        int32_t* list = reinterpret_cast<int32_t*>(hdr + 1);
        DXO_HIP(ctx, hipMemsetAsync(hdr, 0, sizeof(McScratchHeader), s));
        // persistent classify waves (grid-stride) so the batched list append amortises its atomic
        int64_t cgrid = (m + DXO_BLOCK - 1) / DXO_BLOCK;
        const int64_t ccap = (int64_t)ctx->compute_units * 6;
        if (cgrid > ccap) cgrid = ccap;
        const int grid = (int)cgrid;
        hipLaunchKernelGGL(mc_classify, dim3(grid), dim3(DXO_BLOCK), 0, s, L.k, m, deps + off * 4, sigma_n + off * 4,
                           C_tang + off * 16, sigma + off * 4, niter ? niter + off : nullptr,
                           yielding ? yielding + off : nullptr, norm_res ? norm_res + off : nullptr,
                           dlambda ? dlambda + off : nullptr, hdr, list);
        int64_t newton_blocks = (int64_t)ctx->compute_units * ctx->mc_blocks_per_cu;
        const int64_t enough = (m + DXO_BLOCK - 1) / DXO_BLOCK;
        if (newton_blocks > enough) newton_blocks = enough;
#define DXO_MC_NEWTON(W_, S_)                                                                                                  \
    hipLaunchKernelGGL((mc_newton<W_, S_>), dim3((int)newton_blocks), dim3(DXO_BLOCK), 0, s, L.k, deps + off * 4, sigma_n + off * 4, \
                       C_tang + off * 16, sigma + off * 4, niter ? niter + off : nullptr, norm_res ? norm_res + off : nullptr,  \
                       dlambda ? dlambda + off : nullptr, hdr, list)
        const bool same = L.k.same_angle != 0;
        if (ctx->mc_waves_per_simd >= 2) { if (same) DXO_MC_NEWTON(2, true); else DXO_MC_NEWTON(2, false); }
        else { if (same) DXO_MC_NEWTON(1, true); else DXO_MC_NEWTON(1, false); }
#undef DXO_MC_NEWTON
    }
    return DXO_OK;
}

// ------------------------------------------------------------------ inner-Newton summary on the device
// The reference prints, at every call, the unique iteration counts with their multiplicities, max f and max
// residual (demo_plasticity_mohr_coulomb.py:584-591). For device-resident diagnostics this kernel produces the
// same numbers without moving 28 B/point over PCIe: wave-level __shfl_xor max reductions, a per-workgroup LDS
// histogram, then one global atomic per non-empty bin / per maximum and workgroup.
constexpr int MC_HIST_BINS = 1024;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));   // fmax drops NaN like jnp.max does not —
    return v;                                                                  // NaNs are counted separately below
}

__device__ __forceinline__ void atomic_max_double(double* addr, double v) {
    // order-preserving map of IEEE doubles onto unsigned integers
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    bits = (bits >> 63) ? ~bits : (bits | 0x8000000000000000ull);
    atomicMax(reinterpret_cast<unsigned long long*>(addr), bits);
}

__global__ __launch_bounds__(DXO_BLOCK) void mc_summary(int64_t n, int nbins, const int32_t* __restrict__ niter,
                                                        const double* __restrict__ yielding,
                                                        const double* __restrict__ norm_res,
                                                        unsigned long long* __restrict__ hist,
                                                        double* __restrict__ maxima /* [2] encoded, [2] NaN counts */) {
    __shared__ unsigned int lh[MC_HIST_BINS];
    for (int b = threadIdx.x; b < nbins; b += DXO_BLOCK) lh[b] = 0u;
    __syncthreads();
    double my = -(double)INFINITY, mr = -(double)INFINITY;
    unsigned int nan_y = 0, nan_r = 0;
    const int64_t stride = (int64_t)gridDim.x * DXO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * DXO_BLOCK + threadIdx.x; i < n; i += stride) {
        int it = niter[i];
        it = it < 0 ? 0 : (it >= nbins ? nbins - 1 : it);
        atomicAdd(&lh[it], 1u);
        const double y = yielding ? yielding[i] : -(double)INFINITY, r = norm_res ? norm_res[i] : -(double)INFINITY;
        if (y != y) ++nan_y; else my = fmax(my, y);
        if (r != r) ++nan_r; else mr = fmax(mr, r);
    }
    my = wave_max(my);
    mr = wave_max(mr);
    if ((threadIdx.x & 63) == 0) {
        atomic_max_double(maxima + 0, my);
        atomic_max_double(maxima + 1, mr);
    }
    if (nan_y) atomicAdd(reinterpret_cast<unsigned long long*>(maxima + 2), (unsigned long long)nan_y);
    if (nan_r) atomicAdd(reinterpret_cast<unsigned long long*>(maxima + 3), (unsigned long long)nan_r);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += DXO_BLOCK)
        if (lh[b]) atomicAdd(&hist[b], (unsigned long long)lh[b]);
}

int mc_chunk(dxo_ctx* ctx, void* user, int64_t m, void* const* d_in, void* const* d_out, hipStream_t s) {
    const McLaunch& L = *static_cast<const McLaunch*>(user);
    int o = 2;
    int32_t* it = L.d_niter ? (int32_t*)d_out[o++] : nullptr;
    double* yl = L.d_yield ? (double*)d_out[o++] : nullptr;
    double* nr = L.d_res ? (double*)d_out[o++] : nullptr;
    double* dl = L.d_dl ? (double*)d_out[o++] : nullptr;
    return mc_launch(ctx, L, m, (const double*)d_in[0], (const double*)d_in[1], (double*)d_out[0], (double*)d_out[1], it,
                     yl, nr, dl, s);
}

}  // namespace

extern "C" int dxo_mohr_coulomb(dxo_ctx* ctx, const dxo_mc_params* prm, int64_t n, int mem, const double* deps,
                                const double* sigma_n, double* C_tang, double* sigma, int32_t* niter, double* yielding,
                                double* norm_res, double* dlambda) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb: params is NULL");
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb: n < 0");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mohr_coulomb: bad mem");
    if (n > 0 && (!deps || !sigma_n || !C_tang || !sigma)) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb: NULL array");
    if (prm->nitermax < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb: nitermax < 0");
    const uintptr_t a16 = (uintptr_t)deps | (uintptr_t)sigma_n | (uintptr_t)C_tang | (uintptr_t)sigma;
    const uintptr_t a8 = (uintptr_t)yielding | (uintptr_t)norm_res | (uintptr_t)dlambda;
    if (mem == DXO_MEM_DEVICE && (a16 & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb: deps/sigma_n/C_tang/sigma must be 16-byte aligned");
    if ((a16 & 7u) || (a8 & 7u) || ((uintptr_t)niter & 3u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb: misaligned array");
    McLaunch L{mc::make_const(prm->E, prm->nu, prm->c, prm->phi, prm->psi, prm->theta_T, prm->a, prm->tol, prm->nitermax),
               niter != nullptr, yielding != nullptr, norm_res != nullptr, dlambda != nullptr};
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = mc_launch(ctx, L, n, deps, sigma_n, C_tang, sigma, niter, yielding, norm_res, dlambda, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t sd = sizeof(double);
    std::vector<dxo_span> in = {{deps, nullptr, 4 * sd}, {sigma_n, nullptr, 4 * sd}};
    std::vector<dxo_span> out = {{nullptr, C_tang, 16 * sd}, {nullptr, sigma, 4 * sd}};
    if (niter) out.push_back({nullptr, niter, sizeof(int32_t)});
    if (yielding) out.push_back({nullptr, yielding, sd});
    if (norm_res) out.push_back({nullptr, norm_res, sd});
    if (dlambda) out.push_back({nullptr, dlambda, sd});
    return dxo_run_host_pipeline(ctx, n, in, out, mc_chunk, &L);
}

// internal: the kernels on device pointers and an explicit stream (field_ops.hip)
int dxo_mc_launch_device(dxo_ctx* ctx, const dxo_mc_params* prm, int64_t n, const double* deps, const double* sigma_n,
                         double* C_tang, double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda,
                         hipStream_t s) {
    McLaunch L{mc::make_const(prm->E, prm->nu, prm->c, prm->phi, prm->psi, prm->theta_T, prm->a, prm->tol, prm->nitermax),
               niter != nullptr, yielding != nullptr, norm_res != nullptr, dlambda != nullptr};
    return mc_launch(ctx, L, n, deps, sigma_n, C_tang, sigma, niter, yielding, norm_res, dlambda, s);
}

// ------------------------------------------------------------------ history variable resident on the device
// The reference's callback re-reads sigma_n from a closure-captured host array at every call
// (demo_plasticity_mohr_coulomb.py:579) although it changes only at the end of a load step (:728,
// `sigma_n.x.array[:] = sigma.ref_coefficient.x.array`). A dxo_mc_state is its device mirror plus the stress of the last
// call; the commit is that assignment on the device.
extern "C" int dxo_mc_state_create(dxo_ctx* ctx, int64_t n, dxo_mc_state** out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!out) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_create: out is NULL");
    *out = nullptr;
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mc_state_create: n < 0");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bs = ((size_t)n * 4 * sizeof(double) + 255) / 256 * 256;
    void* blob = nullptr;
    DXO_HIP(ctx, hipMalloc(&blob, 2 * bs + 256));
    dxo_mc_state* st = new dxo_mc_state;
    st->n = n;
    st->blob = blob;
    st->sigma_n = reinterpret_cast<double*>(blob);
    st->sigma = reinterpret_cast<double*>(static_cast<char*>(blob) + bs);
    *out = st;
    return DXO_OK;
}

extern "C" void dxo_mc_state_destroy(dxo_ctx* ctx, dxo_mc_state* st) {
    if (!st) return;
    if (ctx) {
        DXO_LOCK(ctx);
        (void)hipSetDevice(ctx->device);
        (void)dxo_ctx_synchronize(ctx);
        if (st->blob) (void)hipFree(st->blob);
    } else if (st->blob) {
        (void)hipFree(st->blob);
    }
    delete st;
}

extern "C" int dxo_mc_state_upload(dxo_ctx* ctx, dxo_mc_state* st, int mem, const double* sigma_n) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_upload: state is NULL");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mc_state_upload: bad mem");
    if (st->n > 0 && !sigma_n) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_upload: NULL array");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    if (st->n > 0) {
        DXO_HIP(ctx, hipMemcpyAsync(st->sigma_n, sigma_n, (size_t)st->n * 4 * sizeof(double),
                                    mem == DXO_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));   // the next call may read the mirror from any of the pipeline's streams
    }
    st->uploaded = true;
    st->has_result = false;
    return DXO_OK;
}

extern "C" int dxo_mc_state_download(dxo_ctx* ctx, dxo_mc_state* st, int mem, double* sigma_n) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_download: state is NULL");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mc_state_download: bad mem");
    if (!st->uploaded) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mc_state_download: nothing has been uploaded");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    if (st->n > 0 && sigma_n) {
        DXO_HIP(ctx, hipMemcpyAsync(sigma_n, st->sigma_n, (size_t)st->n * 4 * sizeof(double),
                                    mem == DXO_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));
    }
    return DXO_OK;
}

extern "C" int dxo_mc_state_pointers(dxo_ctx* ctx, dxo_mc_state* st, double** sigma_n, double** sigma) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_pointers: state is NULL");
    if (sigma_n) *sigma_n = st->sigma_n;
    if (sigma) *sigma = st->sigma;
    return DXO_OK;
}

extern "C" int dxo_mc_state_commit(dxo_ctx* ctx, dxo_mc_state* st) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!st) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_state_commit: state is NULL");
    if (st->n == 0) return DXO_OK;   // an empty partition has nothing to update
    if (!st->has_result)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_mc_state_commit: no call since the last upload / commit — nothing to commit");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    // sigma_n <- sigma (:728): a device-to-device copy at HBM speed (64 B per point)
    DXO_HIP(ctx, hipMemcpyAsync(st->sigma_n, st->sigma, (size_t)st->n * 4 * sizeof(double), hipMemcpyDeviceToDevice, s));
    DXO_HIP(ctx, hipStreamSynchronize(s));   // the pipeline's streams read the mirror next
    st->has_result = false;
    return DXO_OK;
}

extern "C" int dxo_mohr_coulomb_state(dxo_ctx* ctx, const dxo_mc_params* prm, dxo_mc_state* st, int mem, const double* deps,
                                      double* C_tang, double* sigma, int32_t* niter, double* yielding, double* norm_res,
                                      double* dlambda) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !st) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb_state: NULL params or state");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mohr_coulomb_state: bad mem");
    if (!st->uploaded) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb_state: dxo_mc_state_upload has not been called");
    if (prm->nitermax < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb_state: nitermax < 0");
    const int64_t n = st->n;
    if (n == 0) return DXO_OK;
    if (!deps || !C_tang) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb_state: NULL array");
    const uintptr_t a16 = (uintptr_t)deps | (uintptr_t)C_tang | (uintptr_t)sigma;
    const uintptr_t a8 = (uintptr_t)yielding | (uintptr_t)norm_res | (uintptr_t)dlambda;
    if (mem == DXO_MEM_DEVICE && (a16 & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb_state: deps/C_tang/sigma must be 16-byte aligned");
    if ((a16 & 7u) || (a8 & 7u) || ((uintptr_t)niter & 3u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb_state: misaligned array");
    McLaunch L{mc::make_const(prm->E, prm->nu, prm->c, prm->phi, prm->psi, prm->theta_T, prm->a, prm->tol, prm->nitermax),
               niter != nullptr, yielding != nullptr, norm_res != nullptr, dlambda != nullptr};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    st->has_result = false;
    const size_t sd = sizeof(double);
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = mc_launch(ctx, L, n, deps, st->sigma_n, C_tang, st->sigma, niter, yielding, norm_res, dlambda, s);
        if (rc != DXO_OK) return rc;
        if (sigma) DXO_HIP(ctx, hipMemcpyAsync(sigma, st->sigma, (size_t)n * 4 * sd, hipMemcpyDeviceToDevice, s));
        rc = dxo_device_end(ctx, s);
        if (rc == DXO_OK) st->has_result = true;
        return rc;
    }
    if (!sigma) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb_state: host sigma is required");
    std::vector<dxo_span> in = {{deps, nullptr, 4 * sd}, {nullptr, nullptr, 4 * sd, st->sigma_n}};
    std::vector<dxo_span> out = {{nullptr, C_tang, 16 * sd}, {nullptr, sigma, 4 * sd, st->sigma}};
    if (niter) out.push_back({nullptr, niter, sizeof(int32_t)});
    if (yielding) out.push_back({nullptr, yielding, sd});
    if (norm_res) out.push_back({nullptr, norm_res, sd});
    if (dlambda) out.push_back({nullptr, dlambda, sd});
    const int rc = dxo_run_host_pipeline(ctx, n, in, out, mc_chunk, &L);
    if (rc == DXO_OK) st->has_result = true;
    return rc;
}

extern "C" int dxo_mc_summary(dxo_ctx* ctx, int64_t n, const int32_t* niter, const double* yielding, const double* norm_res,
                              int nbins, int64_t* hist, double* max_yielding, double* max_norm_res, int64_t* nan_counts) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (n < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mc_summary: n < 0");
    if (nbins < 1 || nbins > MC_HIST_BINS) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mc_summary: nbins must be in [1, 1024]");
    if (!hist || (n > 0 && !niter)) return dxo_fail(ctx, DXO_E_NULL, "dxo_mc_summary: NULL array");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = dxo_launch_stream(ctx);
    const size_t bytes = (size_t)nbins * sizeof(unsigned long long) + 4 * sizeof(double);
    void* scratch = mc_scratch(ctx, s, bytes + 256);
    if (!scratch) return dxo_hip_fail(ctx, hipErrorOutOfMemory, "dxo_mc_summary: scratch allocation");
    // the summary block sits behind a 256-byte header so it never aliases a list that mc_newton may still read
    unsigned long long* d_hist = reinterpret_cast<unsigned long long*>(static_cast<char*>(scratch));
    double* d_max = reinterpret_cast<double*>(d_hist + nbins);
    DXO_HIP(ctx, hipStreamSynchronize(s));   // scratch is shared with dxo_mohr_coulomb launches on this stream
    DXO_HIP(ctx, hipMemsetAsync(d_hist, 0, bytes, s));
    if (n > 0) {
        int64_t blocks = (n + DXO_BLOCK - 1) / DXO_BLOCK;
        const int64_t cap = (int64_t)ctx->compute_units * 8;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(mc_summary, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, n, nbins, niter, yielding, norm_res, d_hist, d_max);
        DXO_HIP(ctx, hipGetLastError());
    }
    std::vector<unsigned long long> h((size_t)nbins + 4);
    DXO_HIP(ctx, hipMemcpyAsync(h.data(), d_hist, bytes, hipMemcpyDeviceToHost, s));
    DXO_HIP(ctx, hipStreamSynchronize(s));
    for (int b = 0; b < nbins; ++b) hist[b] = (int64_t)h[b];
    auto decode = [](unsigned long long bits) -> double {
        if (bits == 0ull) return -(double)INFINITY;   // nothing recorded
        bits = (bits >> 63) ? (bits & 0x7fffffffffffffffull) : ~bits;
        double v;
        std::memcpy(&v, &bits, sizeof v);
        return v;
    };
    if (max_yielding) *max_yielding = decode(h[nbins]);
    if (max_norm_res) *max_norm_res = decode(h[nbins + 1]);
    if (nan_counts) {
        nan_counts[0] = (int64_t)h[nbins + 2];
        nan_counts[1] = (int64_t)h[nbins + 3];
    }
    return DXO_OK;
}
