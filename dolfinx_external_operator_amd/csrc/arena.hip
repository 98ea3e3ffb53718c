// arena.hip — placement-calibrated device memory for OUTPUT arrays (dxo_output_alloc / dxo_output_free).
//
// Why: the pointwise kernels are HBM-bound and ~77 % of the von Mises traffic is stores. On MI355X the rate of a
// multi-GB streaming-write sweep depends on WHICH allocation it goes to: about 5.6-6.0 TB/s ("slow"), 6.4-6.5
// ("medium") or 6.9-7.1 TB/s ("fast") for pure stores, 5.1-5.4 / 5.65 / 6.2-6.3 TB/s for the 13 : 43 von Mises mix;
// the class is stable for the life of the allocation and its share differs from box to box (0 of 12 ... 5 of 8 fast).
// What round 2's experiments established (scripts/exp/place_exp*.hip, profiles/r02_place_exp*.txt):
// the class is attached to the buffer's VIRTUAL range, not to its physical memory (the same physical chunks are
// slow mapped at one range and fast at another, and a fast range stays fast for other chunks: swap / hybrid tests
// of place_exp7); XCD<->address affinity, row order inside a tile, grid size, chunk / fragment size, VA alignment
// and second mappings do not change it; address-translation miss counts are identical for a fast and a slow range
// and the memory controller sees LESS write back-pressure on the slow one (profiles/r02_place_pmc.txt), so the
// throttle sits upstream of the memory controller; one physical block remapped at range after range lands in ONE
// class every time (place_exp4 b, place_exp5, place_exp9 S2), while allocations that exist SIDE BY SIDE differ.
// Hence the method that works in practice: make several ordinary allocations side by side, time a streaming-write
// sweep on each, keep the first one above "placement_good_GBps" (else the fastest) and free the rest. A solver
// allocates its persistent coefficient buffers once, so this is a set-up cost (~10 ms per candidate at 3.4 GB).
//
// Tried and dropped: ONE physical block (hipMemCreate) remapped over candidate ranges of an address reservation — no
// extra memory during the search, but every range comes out in the same class (5.8-6.2 TB/s in two bench runs,
// 6.8 in place_exp9 S2), keeping the ranges mapped side by side as aliases made all of them slower (4.9-5.1 TB/s),
// and one remap sequence ended in a GPU memory-access fault on this ROCm build (profiles/r02_place_exp9.txt).
//
// Option "placement_mode": 2 = hipMalloc candidates as above (default), 0 = plain hipMalloc. Blocks below
// "placement_min_bytes" (1 GiB) are plain hipMalloc: a working set that small lives in the 256 MB Infinity Cache /
// L2 and has no placement class.
#include <algorithm>
#include <chrono>

#include "dxo_common.h"

namespace {

__global__ __launch_bounds__(DXO_BLOCK) void arena_write_sweep(int64_t n_tiles, dxo_f64x2* __restrict__ dst) {
    // 16 KiB per wave iteration, rows of 1 KiB in order, persistent grid: the sweep whose rate classifies a range
    const int lane = threadIdx.x & (DXO_WAVE - 1), wave = threadIdx.x >> 6;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
        dxo_f64x2* d = dst + t * (16 * DXO_WAVE);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(dxo_f64x2{0.0, 0.0}, d + k * DXO_WAVE + lane);
    }
}

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

void launch_sweep(dxo_ctx* c, void* p, size_t bytes, hipStream_t s) {
    const int64_t n_tiles = (int64_t)(bytes / 16384);
    if (n_tiles > 0) hipLaunchKernelGGL(arena_write_sweep, dim3(c->compute_units * 16), dim3(DXO_BLOCK), 0, s, n_tiles, (dxo_f64x2*)p);
}

// GB/s of `launches` back-to-back sweeps over [p, p + bytes); 0 on error. The block must have been swept before
// (launch_sweep): the first passes over a fresh allocation run ~20 % slower than its steady state (measured: every
// candidate of a 12-candidate search read 4.8-5.2 TB/s when timed right after hipMalloc, gpurun_out r02c).
double probe_range(dxo_ctx* c, void* p, size_t bytes, hipStream_t s, int launches = 4) {
    if (bytes < 16384) return 0.0;
    if (hipEventRecord(c->ev_start, s) != hipSuccess) return 0.0;
    for (int l = 0; l < launches; ++l) launch_sweep(c, p, bytes, s);
    if (hipEventRecord(c->ev_stop, s) != hipSuccess || hipEventSynchronize(c->ev_stop) != hipSuccess) return 0.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) != hipSuccess || ms <= 0.f) return 0.0;
    return (double)(bytes / 16384) * 16384.0 * launches / (ms * 1e-3) / 1e9;
}

// ordinary allocations, all candidates alive until the choice is made (a freed block would be handed out again).
// Groups of four: allocate, sweep each twice untimed, then time each — so no block is timed in its first passes.
bool alloc_by_candidates(dxo_ctx* c, size_t bytes, dxo_arena_block& blk, hipStream_t s) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    int K = (int)c->placement_candidates;
    if (K > DXO_PLACEMENT_MAX) K = DXO_PLACEMENT_MAX;
    const size_t cap = (size_t)(0.6 * (double)free_b) / (bytes ? bytes : 1);
    if ((size_t)K > cap) K = (int)cap;
    if (K < 1) K = 1;
    std::vector<void*> cand;
    int best = -1;
    double best_bw = -1.0;
    bool good = false, oom = false;
    while ((int)cand.size() < K && !good && !oom) {
        const int first = (int)cand.size();
        for (int k = first; k < K && k < first + 4; ++k) {
            void* p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); oom = true; break; }
            cand.push_back(p);
        }
        for (int rep = 0; rep < 2; ++rep)
            for (int k = first; k < (int)cand.size(); ++k) launch_sweep(c, cand[k], bytes, s);
        if (hipStreamSynchronize(s) != hipSuccess) break;
        for (int k = first; k < (int)cand.size(); ++k) {
            const double bw = probe_range(c, cand[k], bytes, s);
            blk.info.probe_GBps[k] = bw;
            blk.info.candidates = k + 1;
            if (bw > best_bw) { best_bw = bw; best = k; }
            if (bw >= (double)c->placement_good_GBps) { good = true; break; }
        }
    }
    if (best < 0) {
        for (void* p : cand) (void)hipFree(p);
        return false;
    }
    // second look: while many candidates coexist every rate reads low (fast ones ~6.0 instead of ~6.9 TB/s) and on a
    // box whose candidates all look alike (4.9-5.1) the first ranking is noise. Keep the three best, free the rest,
    // and time those three again; the winner of THAT round is kept.
    std::vector<int> order(cand.size());
    for (size_t k = 0; k < cand.size(); ++k) order[k] = (int)k;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return blk.info.probe_GBps[a] > blk.info.probe_GBps[b]; });
    size_t keep = order.size() < 3 ? order.size() : 3;
    (void)hipStreamSynchronize(s);
    for (size_t r = keep; r < order.size(); ++r) {
        (void)hipFree(cand[(size_t)order[r]]);
        cand[(size_t)order[r]] = nullptr;
    }
    double final_bw = blk.info.probe_GBps[best];
    if (!good && order.size() > 1) {
        // one more allocation made now that the others are gone joins the final round: on boxes whose candidates all
        // read alike, a block allocated on its own was 3-10 % faster than the pick of the crowd (three runs)
        void* late = nullptr;
        if ((int)cand.size() < DXO_PLACEMENT_MAX && hipMalloc(&late, bytes) == hipSuccess) {
            launch_sweep(c, late, bytes, s);
            launch_sweep(c, late, bytes, s);
            (void)hipStreamSynchronize(s);
            order.insert(order.begin() + (long)keep, (int)cand.size());
            cand.push_back(late);
            blk.info.probe_GBps[cand.size() - 1] = 0.0;
            ++keep;
        } else {
            (void)hipGetLastError();
        }
        best = -1;
        final_bw = -1.0;
        for (size_t r = 0; r < keep; ++r) {
            const int k = order[r];
            const double bw = probe_range(c, cand[(size_t)k], bytes, s, 6);
            if (k == (int)cand.size() - 1 && late) { blk.info.probe_GBps[k] = bw; blk.info.candidates = (int)cand.size(); }
            if (bw > final_bw) { final_bw = bw; best = k; }
        }
        if (best < 0) best = order[0];
    }
    for (size_t k = 0; k < cand.size(); ++k)
        if ((int)k != best && cand[k]) (void)hipFree(cand[k]);
    blk.ptr = cand[(size_t)best];
    blk.bytes = bytes;
    blk.info.chosen_GBps = final_bw;
    blk.info.mode = 2;
    blk.info.chosen = best;
    return true;
}

}  // namespace

void dxo_arena_release_all(dxo_ctx* c) {
    for (auto& b : c->arena) {
        if (b.ptr) (void)hipFree(b.ptr);
    }
    c->arena.clear();
}

extern "C" int dxo_output_alloc(dxo_ctx* c, int64_t bytes, void** ptr) {
    if (!c || !ptr) return DXO_E_NULL;
    DXO_LOCK(c);
    *ptr = nullptr;
    if (bytes < 0) return dxo_fail(c, DXO_E_SIZE, "dxo_output_alloc: negative size");
    DXO_HIP(c, hipSetDevice(c->device));
    const size_t need = bytes > 0 ? (size_t)bytes : 1;
    dxo_arena_block blk;
    std::memset(&blk.info, 0, sizeof blk.info);
    blk.info.chosen = -1;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = c->stream;   // calibration runs on the library's own stream and is synchronous
    bool done = false;
    if ((int64_t)need >= c->placement_min_bytes && c->placement_candidates > 1) {
        if (c->placement_mode >= 1) done = alloc_by_candidates(c, need, blk, s);
    }
    if (!done) {
        std::memset(&blk.info, 0, sizeof blk.info);
        blk.info.chosen = -1;
        DXO_HIP(c, hipMalloc(&blk.ptr, need));
        blk.bytes = need;
    }
    DXO_HIP(c, hipStreamSynchronize(s));
    blk.info.calibration_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    c->arena.push_back(blk);
    *ptr = blk.ptr;
    return DXO_OK;
}

extern "C" int dxo_output_free(dxo_ctx* c, void* ptr) {
    if (!c) return DXO_E_NULL;
    DXO_LOCK(c);
    if (!ptr) return DXO_OK;
    for (size_t i = 0; i < c->arena.size(); ++i) {
        if (c->arena[i].ptr != ptr) continue;
        dxo_arena_block b = c->arena[i];
        c->arena.erase(c->arena.begin() + (long)i);
        DXO_HIP(c, hipSetDevice(c->device));
        DXO_HIP(c, hipDeviceSynchronize());
        DXO_HIP(c, hipFree(b.ptr));
        return DXO_OK;
    }
    return dxo_fail(c, DXO_E_NULL, "dxo_output_free: pointer was not returned by dxo_output_alloc on this context");
}

extern "C" int dxo_output_info(dxo_ctx* c, const void* ptr, dxo_placement_info* info) {
    if (!c || !info) return DXO_E_NULL;
    DXO_LOCK(c);
    for (const auto& b : c->arena)
        if (b.ptr == ptr) {
            *info = b.info;
            return DXO_OK;
        }
    return dxo_fail(c, DXO_E_NULL, "dxo_output_info: pointer was not returned by dxo_output_alloc on this context");
}
