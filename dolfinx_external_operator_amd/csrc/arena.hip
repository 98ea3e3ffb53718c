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
// Round 2, later: the candidates need not all be hipMalloc blocks — see "candidates" below (option "placement_vmm").
//
// Option "placement_mode": 2 = candidates as above (default), 0 = plain hipMalloc. Blocks below
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

// The sweep of the constitutive kernels themselves (von Mises d = 6 proportions): per 64-point tile 13 KiB read from
// three input streams (6 : 6 : 1 rows of 1 KiB) and 43 KiB written to three output streams (36 : 6 : 1) laid out one
// after the other in the candidate, as dxo_von_mises' callers lay out (C_tang, sigma, dp). A block can be fast for ONE
// store stream and ordinary for this pattern: blocks backed by 2 MB chunks topped the single-stream ranking
// (6.3-6.6 TB/s) and then ran vm_tile at 5.6-6.0 TB/s, below hipMalloc blocks that had probed lower — six concurrent
// streams need six times the address-translation reach. So the candidates are ranked with this sweep.
__global__ __launch_bounds__(DXO_BLOCK) void arena_mix_sweep(int64_t n_tiles, const dxo_f64x2* __restrict__ src, dxo_f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & (DXO_WAVE - 1), wave = threadIdx.x >> 6;
    const dxo_f64x2 *i0 = src, *i1 = src + n_tiles * (6 * DXO_WAVE), *i2 = src + n_tiles * (12 * DXO_WAVE);
    dxo_f64x2 *o0 = dst, *o1 = dst + n_tiles * (36 * DXO_WAVE), *o2 = dst + n_tiles * (42 * DXO_WAVE);
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (int64_t)gridDim.x * 4) {
        dxo_f64x2 a = {0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 6; ++k) a += i0[t * (6 * DXO_WAVE) + k * DXO_WAVE + lane] + i1[t * (6 * DXO_WAVE) + k * DXO_WAVE + lane];
        a += i2[t * DXO_WAVE + lane];
        __builtin_nontemporal_store(a, o2 + t * DXO_WAVE + lane);
#pragma unroll
        for (int k = 0; k < 6; ++k) __builtin_nontemporal_store(a, o1 + t * (6 * DXO_WAVE) + k * DXO_WAVE + lane);
#pragma unroll 6
        for (int k = 0; k < 36; ++k) __builtin_nontemporal_store(a, o0 + t * (36 * DXO_WAVE) + k * DXO_WAVE + lane);
    }
}

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// GB/s of `launches` back-to-back sweeps of `shape` over the block; 0 on error. The block must have been swept before:
// the first passes over a fresh allocation run ~20 % slower than its steady state (measured: every candidate of a
// 12-candidate search read 4.8-5.2 TB/s when timed right after hipMalloc, gpurun_out r02c).
// (events of its own, per context = per device: a probe may call the library's entry points, which record the context's
// timing events)
bool cal_events(dxo_ctx* c) {
    for (auto& e : c->cal_ev)
        if (!e && hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; return false; }
    return true;
}
double time_shape(dxo_ctx* c, const dxo_arena_probe& pr, void* p, int shape, hipStream_t s, int launches) {
    hipEvent_t* cal_ev = c->cal_ev;
    if (!cal_events(c) || hipEventRecord(cal_ev[0], s) != hipSuccess) return 0.0;
    for (int l = 0; l < launches; ++l) pr.launch(p, shape, s);
    if (hipEventRecord(cal_ev[1], s) != hipSuccess || hipEventSynchronize(cal_ev[1]) != hipSuccess) return 0.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, cal_ev[0], cal_ev[1]) != hipSuccess || ms <= 0.f) return 0.0;
    return pr.bytes_per_launch * launches / (ms * 1e-3) / 1e9;
}

// best rate over the probe's launch shapes; *shape_out = the shape that gave it
double probe_block(dxo_ctx* c, const dxo_arena_probe& pr, void* p, hipStream_t s, int launches, int* shape_out) {
    double best = 0.0;
    int best_shape = pr.shapes.empty() ? 0 : pr.shapes[0];
    for (int sh : pr.shapes) {
        const double bw = time_shape(c, pr, p, sh, s, launches);
        if (bw > best) { best = bw; best_shape = sh; }
    }
    if (shape_out) *shape_out = best_shape;
    return best;
}

void warm_block(const dxo_arena_probe& pr, void* p, hipStream_t s) {
    for (int sh : pr.shapes) pr.launch(p, sh, s);
    if (pr.shapes.size() < 2) pr.launch(p, pr.shapes.empty() ? 0 : pr.shapes[0], s);
}

// ---- candidates. Two kinds (option "placement_vmm", default on: three of every four candidates are of the second kind):
//   hipMalloc            one driver allocation;
//   2 MB chunks          a virtual range (hipMemAddressReserve) backed by 2 MB physical chunks (hipMemCreate), each
//                        mapped ONCE (hipMemMap). Over ~90 slabs on nine boxes (scripts/exp/place_exp11.hip,
//                        place_exp12.hip) such slabs came out in the fast class about twice as often as hipMalloc
//                        blocks (53-77 % against 25-37 %; physically contiguous hipDeviceMallocContiguous blocks: never),
//                        also on boxes where no hipMalloc candidate did. Building one takes 50-200 ms per 3.4 GB. Teardown
//                        is hipMemUnmap + hipMemRelease + hipMemAddressFree; nothing is ever mapped twice (the remap
//                        sequences of place_exp9 are what faulted), place_exp12 cycles build / partial teardown /
//                        reuse without a fault and with the survivor's data intact.
struct VmmBacking {
    size_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> chunks;
};
constexpr size_t VMM_CHUNK = (size_t)2 << 20;

struct Cand {
    void* p = nullptr;
    VmmBacking* vmm = nullptr;
};

void cand_free(Cand& c) {
    if (!c.p) return;
    if (c.vmm) {
        (void)hipMemUnmap(c.p, c.vmm->va_bytes);
        for (auto& h : c.vmm->chunks) (void)hipMemRelease(h);
        (void)hipMemAddressFree(c.p, c.vmm->va_bytes);
        delete c.vmm;
    } else {
        (void)hipFree(c.p);
    }
    c = Cand();
}

bool cand_alloc(dxo_ctx* c, size_t bytes, bool vmm, Cand& out) {
    out = Cand();
    if (!vmm) {
        if (hipMalloc(&out.p, bytes) != hipSuccess) { (void)hipGetLastError(); out.p = nullptr; return false; }
        return true;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    VmmBacking* b = new VmmBacking();
    const size_t nch = (bytes + VMM_CHUNK - 1) / VMM_CHUNK;
    b->va_bytes = nch * VMM_CHUNK;
    void* va = nullptr;
    bool ok = hipMemAddressReserve(&va, b->va_bytes, VMM_CHUNK, nullptr, 0) == hipSuccess;
    size_t mapped = 0;
    if (ok) {
        b->chunks.reserve(nch);
        for (size_t i = 0; i < nch && ok; ++i) {
            hipMemGenericAllocationHandle_t h;
            ok = hipMemCreate(&h, VMM_CHUNK, &prop, 0) == hipSuccess;
            if (!ok) break;
            b->chunks.push_back(h);
            ok = hipMemMap((char*)va + i * VMM_CHUNK, VMM_CHUNK, 0, h, 0) == hipSuccess;
            if (ok) ++mapped;
        }
        if (ok) ok = hipMemSetAccess(va, b->va_bytes, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        if (va) {
            if (mapped) (void)hipMemUnmap(va, mapped * VMM_CHUNK);
            for (auto& h : b->chunks) (void)hipMemRelease(h);
            (void)hipMemAddressFree(va, b->va_bytes);
        }
        delete b;
        return false;
    }
    out.p = va;
    out.vmm = b;
    return true;
}

// all candidates alive until the choice is made (a freed block would be handed out again).
// Groups of four: allocate, sweep each twice untimed, then time each — so no block is timed in its first passes.
bool search_once(dxo_ctx* c, size_t bytes, const dxo_arena_probe& pr, dxo_arena_block& blk, hipStream_t s) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    int K = (int)c->placement_candidates;
    if (K > DXO_PLACEMENT_MAX) K = DXO_PLACEMENT_MAX;
    const size_t cap = (size_t)(0.6 * (double)free_b) / (bytes ? bytes : 1);
    if ((size_t)K > cap) K = (int)cap;
    if (K < 1) K = 1;
    bool use_vmm = c->placement_vmm != 0;
    blk.info.probe_kind = (int16_t)pr.kind;
    const double good_GBps = pr.good_GBps > 0.0 ? pr.good_GBps : 1e30;
    int tuned[DXO_PLACEMENT_MAX];
    for (int& t : tuned) t = 0;
    std::vector<Cand> cand;
    int best = -1;
    double best_bw = -1.0;
    bool good = false, oom = false;
    auto add = [&](bool vmm) -> bool {
        Cand cd;
        if (vmm && !cand_alloc(c, bytes, true, cd)) { use_vmm = false; vmm = false; }   // no virtual-memory API here: hipMalloc only
        if (!vmm && !cand_alloc(c, bytes, false, cd)) return false;
        if (cd.vmm) blk.info.vmm_mask |= 1u << cand.size();
        cand.push_back(cd);
        return true;
    };
    while ((int)cand.size() < K && !good && !oom) {
        const int first = (int)cand.size();
        for (int k = first; k < K && k < first + 4; ++k)
            if (!add(use_vmm && ((k & 3) != 0 || c->placement_vmm >= 2))) { oom = true; break; }   // one hipMalloc block, three of 2 MB chunks, per group of four (placement_vmm 2: chunks only)
        for (int k = first; k < (int)cand.size(); ++k) warm_block(pr, cand[(size_t)k].p, s);
        if (hipStreamSynchronize(s) != hipSuccess) break;
        for (int k = first; k < (int)cand.size(); ++k) {
            const double bw = probe_block(c, pr, cand[(size_t)k].p, s, 4, &tuned[k]);
            blk.info.probe_GBps[k] = bw;
            blk.info.candidates = k + 1;
            if (bw > best_bw) { best_bw = bw; best = k; }
            if (bw >= good_GBps) { good = true; break; }
        }
    }
    if (best < 0) {
        for (auto& cd : cand) cand_free(cd);
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
    for (size_t r = keep; r < order.size(); ++r) cand_free(cand[(size_t)order[r]]);
    double final_bw = blk.info.probe_GBps[best];
    if (!good && order.size() > 1) {
        // one more allocation made now that the others are gone joins the final round: on boxes whose candidates all
        // read alike, a block allocated on its own was 3-10 % faster than the pick of the crowd (three runs)
        bool late = false;
        if ((int)cand.size() < DXO_PLACEMENT_MAX && add(use_vmm)) {
            late = true;
            warm_block(pr, cand.back().p, s);
            (void)hipStreamSynchronize(s);
            order.insert(order.begin() + (long)keep, (int)cand.size() - 1);
            blk.info.probe_GBps[cand.size() - 1] = 0.0;
            ++keep;
        }
        best = -1;
        final_bw = -1.0;
        for (size_t r = 0; r < keep; ++r) {
            const int k = order[r];
            const double bw = probe_block(c, pr, cand[(size_t)k].p, s, 6, &tuned[k]);
            if (k == (int)cand.size() - 1 && late) { blk.info.probe_GBps[k] = bw; blk.info.candidates = (int)cand.size(); }
            if (bw > final_bw) { final_bw = bw; best = k; }
        }
        if (best < 0) best = order[0];
    }
    (void)hipStreamSynchronize(s);
    for (size_t k = 0; k < cand.size(); ++k)
        if ((int)k != best) cand_free(cand[k]);
    // Round 6: the rate that counts is the one the block has AFTER its rivals are gone. On two leases of five a chunk-backed winner that had
    // probed at 6.3 TB/s ran the caller's launches at 4.7-5.7 TB/s afterwards (profiles/r06_bench_by_lease.txt); unmapping the other ranges
    // is the one thing that happens in between. The block is timed once more here, alone, and THAT rate is what the record, the
    // accept test of alloc_by_candidates and dxo_placement_info report (a drop buys the next search).
    {
        int sh = tuned[best];
        const double post = probe_block(c, pr, cand[(size_t)best].p, s, 6, &sh);
        (void)hipStreamSynchronize(s);
        if (post > 0.0) {
            if (post < 0.97 * final_bw) tuned[best] = sh;      // the shape that is best NOW
            final_bw = post;
        }
    }
    blk.ptr = cand[(size_t)best].p;
    blk.vmm = cand[(size_t)best].vmm;
    blk.bytes = bytes;
    blk.info.chosen_GBps = final_bw;
    blk.info.mode = 2;
    blk.info.chosen = best;
    blk.info.tuned_blocks_per_cu = tuned[best];
    return true;
}

void block_free(dxo_arena_block& b) {
    Cand cd;
    cd.p = b.ptr;
    cd.vmm = static_cast<VmmBacking*>(b.vmm);
    cand_free(cd);
    b.ptr = nullptr;
    b.vmm = nullptr;
}

// ---- a search may come up empty: the share of fast ranges differs from box to box and from moment to moment (round 5: the
// factory's calibration kept a 5.7 TB/s block in 2 runs of 5 while the same context had just found a 6.3 TB/s one for another
// operator). Two tests decide whether the winner of a search is a block of the fast class, and a failed test buys ONE MORE
// search with fresh candidates — made while the winner so far stays allocated, so the new candidates are other ranges:
//   (i)  the context remembers the best rate a calibration of this (probe kind, block size) ever kept: a winner below
//        0.97 of it is rejected (option "placement_accept_pct", 97);
//   (ii) without a record, a winner that does not stand out from its own crowd (below 1.06 x the median candidate, option
//        "placement_standout_pct", 106: a fast block among ordinary ones reads 1.15-1.2 x) is rejected ONCE — unless it runs at the
//        rate fast blocks are known to reach ("placement_good_mix_GBps": on a box whose candidates are all fast nothing stands out).
// Bounded by option "placement_rounds" (3 searches at most); the winner of every search meets the winner so far head to head
// (same number of blocks alive for both timings) and the slower one is freed. info.rounds reports the searches made.
int64_t class_key(int kind, size_t bytes) {
    int lg = 0;
    for (size_t b = bytes; b > 1; b >>= 1) ++lg;
    const int half = (bytes >> (lg > 0 ? lg - 1 : 0)) & 1;     // two classes per octave
    return (int64_t)kind * 256 + lg * 2 + half;
}

double crowd_median(const dxo_placement_info& info) {
    std::vector<double> v;
    for (int k = 0; k < info.candidates && k < DXO_PLACEMENT_MAX; ++k)
        if (info.probe_GBps[k] > 0.0) v.push_back(info.probe_GBps[k]);
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

// ---- the retained block (round 6). A calibration costs 2-7 s and, on some boxes, fast ranges are rare: one lease showed ONE fast block among 68
// candidates, so an operator created after another one of the same size had been deleted could not find what its predecessor had just given
// back. dxo_output_free therefore keeps ONE calibrated block (the fastest it has been handed) instead of releasing it, and the next request of
// exactly its size and probe kind is given that block — re-timed first, and taken only if it still runs at placement_accept_pct of the class
// record. It is released when a search starts (it must not sit among the candidates), when dxo_device_alloc would otherwise fail, and when the
// context closes. Option "placement_cache" = 0 turns it off. info.rounds = 0 marks a block that came from here.
bool cache_take(dxo_ctx* c, size_t bytes, const dxo_arena_probe& pr, dxo_arena_block& blk, hipStream_t s) {
    for (size_t i = 0; i < c->arena_cache.size(); ++i) {
        dxo_arena_block& e = c->arena_cache[i];
        if (e.bytes != bytes || e.info.probe_kind != pr.kind) continue;
        // the request's rule about chunk-backed ranges holds for a retained block too (placement_vmm = 0: RCCL buffers must be plain hipMalloc blocks)
        if ((c->placement_vmm == 0 && e.vmm) || (c->placement_vmm >= 2 && !e.vmm)) continue;
        int sh = e.info.tuned_blocks_per_cu;
        const double bw = probe_block(c, pr, e.ptr, s, 6, &sh);
        (void)hipStreamSynchronize(s);
        const int64_t key = class_key(pr.kind, bytes);
        const double best_seen = c->placement_best.count(key) ? c->placement_best[key] : 0.0;
        if (bw <= 0.0 || bw < 0.01 * (double)c->placement_accept_pct * best_seen) return false;      // no longer what it was: the caller drops it and searches
        blk = e;
        blk.info.chosen_GBps = bw;
        blk.info.tuned_blocks_per_cu = sh;
        blk.info.rounds = 0;
        c->arena_cache.erase(c->arena_cache.begin() + (long)i);
        return true;
    }
    return false;
}

bool alloc_by_candidates(dxo_ctx* c, size_t bytes, const dxo_arena_probe& pr, dxo_arena_block& blk, hipStream_t s) {
    if (c->placement_cache && cache_take(c, bytes, pr, blk, s)) return true;
    dxo_arena_cache_drop(c);
    const int64_t key = class_key(pr.kind, bytes);
    int max_rounds = (int)c->placement_rounds;
    if (max_rounds < 1) max_rounds = 1;
    if (max_rounds > 4) max_rounds = 4;
    dxo_arena_block held;      // winner so far
    bool have = false, standout_retry_used = false;
    int rounds = 0;
    for (int r = 0; r < max_rounds; ++r) {
        dxo_arena_block cur;
        std::memset(&cur.info, 0, sizeof cur.info);
        cur.info.chosen = -1;
        if (!search_once(c, bytes, pr, cur, s)) break;
        ++rounds;
        if (have) {
            // head to head, two blocks alive: the rates both blocks were kept at came from different crowds
            int sh_h = held.info.tuned_blocks_per_cu, sh_c = cur.info.tuned_blocks_per_cu;
            const double bw_h = probe_block(c, pr, held.ptr, s, 6, &sh_h), bw_c = probe_block(c, pr, cur.ptr, s, 6, &sh_c);
            (void)hipStreamSynchronize(s);
            held.info.chosen_GBps = bw_h;
            held.info.tuned_blocks_per_cu = sh_h;
            cur.info.chosen_GBps = bw_c;
            cur.info.tuned_blocks_per_cu = sh_c;
            if (bw_c > bw_h) {
                block_free(held);
                held = cur;
            } else {
                block_free(cur);
            }
        } else {
            held = cur;
            have = true;
        }
        const double best_seen = c->placement_best.count(key) ? c->placement_best[key] : 0.0;
        const double rate = held.info.chosen_GBps;
        if (pr.good_GBps > 0.0 && rate >= pr.good_GBps) break;      // the caller's own "good enough"
        if (best_seen > 0.0) {
            if (rate >= 0.01 * (double)c->placement_accept_pct * best_seen) break;
            continue;
        }
        // a winner at the rate fast blocks are known to reach (option "placement_good_mix_GBps", 6250: 0.78 of the spec peak) needs no
        // second opinion: on a box whose candidates are ALL fast nothing stands out and the retry would only cost its two seconds
        if (rate >= (double)c->placement_good_mix_GBps) break;
        const double med = crowd_median(held.info);
        if (held.info.candidates < 4 || med <= 0.0 || rate >= 0.01 * (double)c->placement_standout_pct * med || standout_retry_used) break;
        standout_retry_used = true;
    }
    if (!have) return false;
    blk = held;
    blk.info.rounds = rounds;
    double& rec = c->placement_best[key];
    if (blk.info.chosen_GBps > rec) rec = blk.info.chosen_GBps;
    return true;
}

}  // namespace

bool dxo_arena_alloc_calibrated(dxo_ctx* c, size_t bytes, const dxo_arena_probe& probe, dxo_arena_block& blk, hipStream_t s) {
    std::memset(&blk.info, 0, sizeof blk.info);
    blk.info.chosen = -1;
    return alloc_by_candidates(c, bytes, probe, blk, s);
}

void dxo_arena_register(dxo_ctx* c, const dxo_arena_block& blk) { c->arena.push_back(blk); }

int dxo_arena_tuned_shape(dxo_ctx* c, const void* ptr) {
    for (const auto& b : c->arena)
        if (b.ptr && (const char*)ptr >= (const char*)b.ptr && (const char*)ptr < (const char*)b.ptr + b.bytes)
            return b.info.mode == 2 && b.info.probe_kind == 2 ? b.info.tuned_blocks_per_cu : -1;   // shapes of vm_tile only
    return -1;
}

// true when `ptr` lies in an arena block built from 2 MB physical chunks (hipMemCreate / hipMemMap): such a range is
// accessible from the owning device only and cannot be exported with hipIpcGetMemHandle, so it must never be handed to RCCL
bool dxo_arena_is_vmm(dxo_ctx* c, const void* ptr) {
    for (const auto& b : c->arena)
        if (b.ptr && (const char*)ptr >= (const char*)b.ptr && (const char*)ptr < (const char*)b.ptr + b.bytes) return b.vmm != nullptr;
    return false;
}

void dxo_arena_cache_drop(dxo_ctx* c) {
    if (c->arena_cache.empty()) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& b : c->arena_cache)
        if (b.ptr) block_free(b);
    c->arena_cache.clear();
}

void dxo_arena_release_all(dxo_ctx* c) {
    dxo_arena_cache_drop(c);
    for (auto& b : c->arena)
        if (b.ptr) block_free(b);
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
    if ((int64_t)need >= c->placement_min_bytes && c->placement_candidates > 1 && c->placement_mode >= 1) {
        // the generic probes: one stream of stores, or the kernels' six-stream read + write sweep (needs an input array)
        dxo_arena_probe pr;
        void* probe_src = nullptr;
        const size_t tiles6 = need / (43 * 1024);
        if (c->placement_probe != 0 && tiles6 > 0 && hipMalloc(&probe_src, tiles6 * 13 * 1024) == hipSuccess &&
            hipMemsetAsync(probe_src, 0, tiles6 * 13 * 1024, s) == hipSuccess) {
            const int grid = c->compute_units * 16;
            pr.launch = [=](void* p, int, hipStream_t st) {
                hipLaunchKernelGGL(arena_mix_sweep, dim3(grid), dim3(DXO_BLOCK), 0, st, (int64_t)tiles6, (const dxo_f64x2*)probe_src, (dxo_f64x2*)p);
            };
            pr.bytes_per_launch = (double)tiles6 * 56.0 * 1024.0;
            pr.good_GBps = (double)c->placement_good_mix_GBps;
            pr.kind = 1;
        } else {
            (void)hipGetLastError();
            if (probe_src) (void)hipFree(probe_src);
            probe_src = nullptr;
            const int64_t tiles = (int64_t)(need / 16384);
            const int grid = c->compute_units * 16;
            pr.launch = [=](void* p, int, hipStream_t st) {
                if (tiles > 0) hipLaunchKernelGGL(arena_write_sweep, dim3(grid), dim3(DXO_BLOCK), 0, st, tiles, (dxo_f64x2*)p);
            };
            pr.bytes_per_launch = (double)tiles * 16384.0;
            pr.good_GBps = (double)c->placement_good_GBps;
            pr.kind = 0;
        }
        done = need >= 16384 && alloc_by_candidates(c, need, pr, blk, s);
        (void)hipStreamSynchronize(s);
        if (probe_src) (void)hipFree(probe_src);
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

// The caller's own consumer as the probe: `launch(block, shape, user)` enqueues ONE pass of the kernel that will write the
// block, on the context's launch stream, and returns without synchronising — typically a call of the dxo_* entry point
// itself with device pointers into `block` (the context's lock is recursive: the callback runs on the calling thread).
extern "C" int dxo_output_alloc_probed(dxo_ctx* c, int64_t bytes, dxo_probe_launch launch, void* user, double bytes_per_launch,
                                       const int32_t* shapes, int n_shapes, void** ptr) {
    if (!c || !ptr) return DXO_E_NULL;
    DXO_LOCK(c);
    *ptr = nullptr;
    if (!launch) return dxo_fail(c, DXO_E_NULL, "dxo_output_alloc_probed: launch is NULL");
    if (bytes < 0 || n_shapes < 0 || n_shapes > 8) return dxo_fail(c, DXO_E_SIZE, "dxo_output_alloc_probed: bad size / more than 8 shapes");
    DXO_HIP(c, hipSetDevice(c->device));
    const size_t need = bytes > 0 ? (size_t)bytes : 1;
    dxo_arena_block blk;
    std::memset(&blk.info, 0, sizeof blk.info);
    blk.info.chosen = -1;
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = dxo_launch_stream(c);   // where the caller's entry points launch
    bool done = false;
    if ((int64_t)need >= c->placement_min_bytes && c->placement_candidates > 1 && c->placement_mode >= 1) {
        dxo_arena_probe pr;
        pr.launch = [=](void* p, int shape, hipStream_t) { launch(p, shape, user); };
        pr.shapes.assign(shapes && n_shapes > 0 ? shapes : nullptr, shapes && n_shapes > 0 ? shapes + n_shapes : nullptr);
        if (pr.shapes.empty()) pr.shapes = {0};
        pr.bytes_per_launch = bytes_per_launch > 0.0 ? bytes_per_launch : (double)need;
        pr.kind = 3;
        done = alloc_by_candidates(c, need, pr, blk, s);
        (void)hipStreamSynchronize(s);
    }
    if (!done) {
        std::memset(&blk.info, 0, sizeof blk.info);
        blk.info.chosen = -1;
        blk.vmm = nullptr;
        DXO_HIP(c, hipMalloc(&blk.ptr, need));
        blk.bytes = need;
    }
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
        if (c->placement_cache && b.info.mode == 2 && b.info.chosen_GBps > 0.0) {
            // retained for the next request of its size (see cache_take): one block at most, the faster of the two
            if (c->arena_cache.empty()) {
                c->arena_cache.push_back(b);
                return DXO_OK;
            }
            if (b.info.chosen_GBps > c->arena_cache[0].info.chosen_GBps) std::swap(b, c->arena_cache[0]);
        }
        block_free(b);
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
