// dxo_common.h — internal declarations shared by the HIP translation units of libdxo_hip.so.
// Not part of the ABI (that is include/dxo.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dxo.h"

typedef double dxo_f64x2 __attribute__((ext_vector_type(2)));

constexpr int DXO_WAVE = 64;          // gfx950 wavefront
constexpr int DXO_BLOCK = 256;        // 4 waves per workgroup
constexpr int DXO_HOST_SLOTS = 3;     // H2D / kernel / D2H pipeline depth

// One block handed out by dxo_output_alloc (arena.hip)
struct dxo_arena_block {
    void* ptr = nullptr;
    size_t bytes = 0;
    dxo_placement_info info;
    void* vmm = nullptr;   // arena.hip: backing of a block built from 2 MB physical chunks (virtual-memory API); NULL = hipMalloc
};
struct dxo_ctx;
void dxo_arena_release_all(dxo_ctx* ctx);
void dxo_arena_cache_drop(dxo_ctx* ctx);      // frees the retained block, if any (a search is about to start, memory ran short, the context closes)

// How dxo_output_alloc's calibration exercises a candidate block (arena.hip). The generic probes are sweeps of
// arena.hip's own; a kernel family can bring its own launch (dxo_vm_output_alloc: vm_tile itself) and a list of launch
// shapes to try — the best shape of the block kept is remembered in its record (dxo_placement_info::tuned_blocks_per_cu).
struct dxo_arena_probe {
    std::function<void(void* block, int shape, hipStream_t s)> launch;   // one sweep over the whole block, asynchronous
    std::vector<int> shapes = {0};
    double bytes_per_launch = 0.0;   // what a launch moves (for the GB/s in the record)
    double good_GBps = 0.0;          // early exit of the search (0: never)
    int kind = 0;                    // dxo_placement_info::probe_kind
};
// bytes: size of the block; true on success (blk filled, registered by the caller)
bool dxo_arena_alloc_calibrated(dxo_ctx* ctx, size_t bytes, const dxo_arena_probe& probe, dxo_arena_block& blk, hipStream_t s);
void dxo_arena_register(dxo_ctx* ctx, const dxo_arena_block& blk);
// tuned launch shape of the arena block that contains `ptr`, or -1
int dxo_arena_tuned_shape(dxo_ctx* ctx, const void* ptr);
// Kernel variants that were measured and not shipped live under scripts/exp/ and are compiled into their translation unit only with
// -DDXO_EXPERIMENTS; dxo_ctx_set_option asks the owning unit whether a value is available in this build (DXO_E_OPTION otherwise).
int dxo_icnn_variant_max();            // icnn.hip: 2 in the product build, 4 with the experiments
bool dxo_adjoint_patch_available();    // adjoint.hip
// does `ptr` lie in an arena block backed by 2 MB physical chunks (not shareable with peers / RCCL)?
bool dxo_arena_is_vmm(dxo_ctx* ctx, const void* ptr);

// Host worker threads of a context (dxo_ctx.hip): the host half of the DXO_MEM_HOST pipeline (tangent rebuild from
// the returned state while later chunks are still on the PCIe link). Created on first use, joined by dxo_ctx_destroy.
struct dxo_host_pool;   // csrc/host_pool.h

struct dxo_ctx {
    int device = 0;
    std::recursive_mutex mu;            // every public entry point holds it: calls on one ctx are serialised
    dxo_host_pool* pool = nullptr;
    hipStream_t stream = nullptr;       // library-owned compute stream
    hipStream_t user_stream = nullptr;  // borrowed (dxo_ctx_set_stream)
    bool use_user_stream = false;
    hipStream_t slot_stream[DXO_HOST_SLOTS] = {nullptr, nullptr, nullptr};
    void* slot_buf[DXO_HOST_SLOTS] = {nullptr, nullptr, nullptr};
    size_t slot_bytes = 0;              // capacity of each slot buffer
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t cal_ev[2] = {nullptr, nullptr};   // arena.hip: the calibration's own pair (a probe may call entry points that record ev_*)
    bool ev_pending = false;            // device-path events recorded, not yet read
    int compute_units = 256;
    // options
    int64_t vm_variant = 1;
    int64_t host_chunk_points = 1 << 18;
    int64_t nontemporal = 1;
    int64_t timing = 0;
    int64_t blocks_per_cu = 0;          // 0: one wave-tile per wave (no grid stride)
    int64_t mc_variant = 2;             // 0: lane = point; 1: classify + compacted Newton with lane refill (two kernels); 2: both in one persistent kernel
    int64_t mc_blocks_per_cu = 3;       // persistent Newton workgroups per CU (2: 1.39 ms, 3: 1.36, 4: 1.36 at 10^7 points)
    int64_t icnn_variant = 2;           // fp32 network: 0 VALU lane-per-point kernel; wave-per-64-points MFMA kernels: 1 fp32-input MFMA, 2 split-bf16 MFMA
    int64_t adjoint_cell = 1;           // virtual work of eps on the standard elements: lane = cell kernel (0: wave-group kernel)
    int64_t operand_cell = 1;           // dxo_eval_operand, eps on the 2-D standard elements: lane = cell kernel (operand_cell.h); 0: wave-group kernel
    int64_t vm_residual_fused = 0;      // dxo_von_mises_residual on Q2 hexahedra: 1 = one kernel (stress scattered from registers, cell8_mfma.h; round 5: 1.11 against
                                        // 1.14-1.16 ms per 10^7 points, profiles/r05_mfma_scatter.txt; R differs from the two calls in the last bits), 0 = field + adjoint calls
    int64_t consumer_overwrite = 0;     // dxo_operand_adjoint / dxo_tangent_apply* / dxo_tangent_diagonal* / dxo_von_mises_residual: 1 = out is SET to the
                                        // assembled vector instead of added to (a Krylov matvec needs no memset and node_sum no read of out)
    int64_t adjoint_patch = 0;          // internal force on hexahedra, whole mesh: 1 = patch form (entries meet in LDS, adjoint_patch.h; measured SLOWER, profiles/r05_patch_form.txt), 0 = element vectors + node sums
    int64_t mgpu_chunks = 4;            // DXO_GATHER_COMPACT_PIPELINED: pieces of a rank's block (kernel of piece k + 1 beside the exchange of piece k)
    int64_t adjoint_mfma = 1;           // Q2 / Q1 hexahedra, 2x2x2 rule: the consumer-side scatter B^T t as f64 MFMAs (c8m_contract in adjoint.hip); 0 = DPP reduce-scatter (cell8_dpp.h)
    int64_t adjoint_atomics = 0;        // adjoint kernels: 1 = fp64 atomics into the dof vector, 0 = element vectors + node sums
    int64_t mc_part_points = (int64_t)1 << 30;   // Mohr-Coulomb: points per classify/Newton pass (int32 list entries)
    int64_t mc_waves_per_simd = 1;      // kept for option compatibility: mc_newton keeps its lane state in LDS (mc_core.h LaneLds) and
                                        // fits two waves per SIMD (240 VGPRs, no scratch) whatever this says
    // output arena (arena.hip)
    int64_t placement_mode = 2;         // 0 plain hipMalloc, 2 (or any value >= 1) hipMalloc candidates
    int64_t placement_candidates = 16;  // allocations / ranges tried at most (<= DXO_PLACEMENT_MAX)
    int64_t placement_min_bytes = (int64_t)1 << 30;
    int64_t placement_vmm = 1;          // three of every four candidates are built from 2 MB physical chunks (arena.hip)
    int64_t placement_probe = 1;        // 1: rank candidates with the six-stream read + write sweep of the kernels, 0: one store stream
    int64_t placement_good_mix_GBps = 6250;   // early-exit rate of the six-stream sweep (algorithmic GB/s)
    int64_t placement_good_GBps = 6800; // stop searching at the first candidate whose write sweep reaches this
    int64_t placement_rounds = 3;       // candidate searches per calibration at most (a rejected winner buys one more, arena.hip)
    int64_t placement_accept_pct = 97;  // a winner below this share of the best rate ever kept for its (probe, size) class is rejected
    int64_t placement_standout_pct = 106;   // without such a record: a winner below this share of its crowd's median is rejected once
    std::map<int64_t, double> placement_best;   // (probe kind, size class) -> best rate a calibration of this context has kept
    std::vector<dxo_arena_block> arena;
    std::vector<dxo_arena_block> arena_cache;   // at most one calibrated block that its owner has freed (arena.hip: handed to the next request of its size)
    int64_t vm_mark_indeterminate = 0;  // DEVICE-path von Mises: dp = -0.0 at f_elastic == 0 exactly (the reference's 0/0, :318), for
                                        // consumers that rebuild the tangent from (sigma, dp); dxo_vm_clear_marks restores +0
    int64_t vm_host_tangent = 0;        // DXO_MEM_HOST von Mises: 0 copy C_tang over PCIe, 1 copy (sigma, dp) and rebuild C_tang on the host
    int64_t host_threads = 32;          // worker threads of the host half of the pipeline (capped by the hardware's)
    int64_t vm_rebuild_chunk_points = 1 << 17;   // pipeline chunk of the vm_host_tangent = 1 mode
    int64_t vm_rebuild_min_points = 1 << 16;     // smaller batches take the copy mode (round 6: the rebuild wins back to back from 15 000 points on — profiles/r06_demo_latency.txt —,
                                                 // but a solver's calls are milliseconds apart and then pay the wake-up of the host threads: 2^16 keeps a margin of 0.1 ms)
    // small-batch path of the host pipeline: one pinned staging buffer, one H2D, one D2H, events made once
    void* small_pinned = nullptr;
    size_t small_pinned_bytes = 0;
    hipEvent_t small_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int64_t host_small_bytes = 8 << 20;   // batches whose inputs + outputs fit this many bytes take the small path
    int64_t host_zero_copy_bytes = 8 << 20;   // ... and below this many bytes the kernel reads / writes page-locked host memory itself (no DMA)
    int64_t host_zero_copy_piece_bytes = 1 << 20;   // ... in pieces of about this many bytes (at most 4): piece k + 1 is packed while piece k runs
    void* scratch[DXO_HOST_SLOTS + 1] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[DXO_HOST_SLOTS + 1] = {0, 0, 0, 0};
    void* stage[DXO_HOST_SLOTS + 1] = {nullptr, nullptr, nullptr, nullptr};   // dxo_stage: operand values in front of a pointwise kernel
    size_t stage_bytes[DXO_HOST_SLOTS + 1] = {0, 0, 0, 0};
    hipStream_t scratch_stream = nullptr;   // stream of the last DEVICE-path launch that used scratch[DXO_HOST_SLOTS]
    bool scratch_stream_set = false;
    int64_t placement_cache = 1;        // dxo_output_free keeps ONE calibrated block for the next request of the same size (0: frees at once)
    int64_t assign_validate = 1;        // dxo_assign: check flat_dofs against coeff_size on the device (one sync per call)
    int64_t assign_plan_form = 0;       // dxo_assign_plan_create: 0 = keep both forms of a large plan and let the first apply time them, 1 = dof order only, 2 = source order
    int64_t assign_owner_bits = 0;      // owner words of the last-writer pass: 0 = 32-bit while the entry count fits, 64 = always wide (what > 2^32 - 2 entries take)
    dxo_timing last = {0, 0, 0, 0};
    std::string err;
};

inline hipStream_t dxo_launch_stream(dxo_ctx* c) { return c->use_user_stream ? c->user_stream : c->stream; }

inline int dxo_fail(dxo_ctx* c, int code, const char* what) {
    if (c) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "%s (code %d)", what, code);
        c->err = buf;
    }
    return code;
}

inline int dxo_hip_fail(dxo_ctx* c, hipError_t e, const char* where) {
    if (c) {
        char buf[512];
        std::snprintf(buf, sizeof buf, "%s: %s (hipError %d)", where, hipGetErrorString(e), (int)e);
        c->err = buf;
    }
    return (int)e > 0 ? (int)e : 999;
}

#define DXO_HIP(ctx, call)                                             \
    do {                                                               \
        hipError_t _e = (call);                                        \
        if (_e != hipSuccess) return dxo_hip_fail((ctx), _e, #call);   \
    } while (0)

// One array of a pointwise map: host/device base pointer and bytes per quadrature point.
struct dxo_span {
    const void* in = nullptr;   // for inputs
    void* out = nullptr;        // for outputs
    size_t bytes_pp = 0;
    // Array that LIVES on the device (dxo_vm_state): an input with `dev` is neither staged nor copied — the chunk's
    // kernel reads dev + first*bytes_pp; an output with `dev` is written there by the kernel and, when `out` is set,
    // copied to the host from there.
    void* dev = nullptr;
};

// Device mirror of the von Mises history variables plus the results of the last call (include/dxo.h, dxo_vm_state_*):
// ONE allocation, [sigma_n n*d | sigma n*d | p n | dp n] doubles, every part on a 256-byte border.
struct dxo_vm_state {
    int d = 0;
    int64_t n = 0;
    void* blob = nullptr;
    double *sigma_n = nullptr, *sigma = nullptr, *p = nullptr, *dp = nullptr;
    bool uploaded = false;     // sigma_n / p hold data
    bool has_result = false;   // sigma / dp hold the results of a call made after the last upload / commit
};

// Launch callback for the chunked host pipeline: device pointers of the chunk, in the same
// order as the spans that were passed, and the number of points in the chunk.
typedef int (*dxo_chunk_launch)(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const* d_in,
                                void* const* d_out, hipStream_t stream);

// Optional host-side completion hook of the pipeline: called on the calling thread once the D2H copies of points
// [first, first + m) have landed, while later chunks are still in flight.
typedef int (*dxo_chunk_post)(dxo_ctx* ctx, void* user, int64_t first, int64_t m);

// Run fn(begin, end) over [0, n) split into contiguous ranges on the context's host threads; returns when all are done.
void dxo_host_parallel_for(dxo_ctx* ctx, int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& fn);
// registry of page-locked host ranges (dxo_ctx.hip): noted by dxo_host_alloc / dxo_host_register, forgotten by their counterparts
void dxo_pinned_note(void* base, size_t bytes, bool add);
void* dxo_pinned_mapped(const void* p, size_t bytes);

// Every extern "C" entry point that takes a ctx starts with this (ctx may be NULL: the guard is then a no-op).
#define DXO_LOCK(ctx) std::unique_lock<std::recursive_mutex> dxo_lock_guard_ = (ctx) ? std::unique_lock<std::recursive_mutex>((ctx)->mu) : std::unique_lock<std::recursive_mutex>()

// H2D -> kernel -> D2H, chunked over points and rotated over DXO_HOST_SLOTS streams so copies
// of one chunk overlap the kernel of another. Blocks until every output byte is on the host.
// `n` counts units of `points_per_unit` quadrature points (1: points; nq: cells — bytes_pp is then per cell); the
// chunk size option host_chunk_points stays in points.
// `stream_once`: the kernel reads every input byte once and writes every output byte once with full-width stores — only then
// may tiny batches run it directly on the device-mapped pinned staging block (option host_zero_copy_bytes); kernels that re-read
// inputs or store partial lines (Mohr-Coulomb's list gather / stash, the network kernels) keep the DMA copies.
int dxo_run_host_pipeline(dxo_ctx* ctx, int64_t n, const std::vector<dxo_span>& inputs,
                          const std::vector<dxo_span>& outputs, dxo_chunk_launch launch, void* user,
                          int64_t points_per_unit = 1, dxo_chunk_post post = nullptr, bool stream_once = false);

// Device-path bracket: optional event timing around a launch sequence.
int dxo_device_begin(dxo_ctx* ctx, hipStream_t s);
int dxo_device_end(dxo_ctx* ctx, hipStream_t s);

int dxo_grid_for_tiles(const dxo_ctx* ctx, int64_t n_tiles, int tiles_per_block);

// ctx-owned device scratch of at least `bytes`: one buffer per host-pipeline slot stream plus ONE for device-path
// launches. The device-path buffer is shared by every caller stream, so when the launch stream changes the previous
// stream is drained first (two user streams must not race on it). nullptr on allocation failure.
void* dxo_scratch(dxo_ctx* ctx, hipStream_t s, size_t bytes);
// A second family of the same kind: the `*_field` entry points of the Newton / network kernels stage the operand values of
// a chunk here while the kernel behind them uses dxo_scratch for its own lists (field_ops.hip).
void* dxo_stage(dxo_ctx* ctx, hipStream_t s, size_t bytes);

// Internal launchers shared between translation units (device pointers, explicit stream, no locking, no event bracket):
struct dxo_mesh;
struct dxo_icnn;
int dxo_operand_launch_range(dxo_ctx* ctx, const dxo_mesh* mesh, int kind, int bs, const double* u_dev, int64_t cell0,
                             int64_t n_cells, double* out_dev, hipStream_t s);
int dxo_mc_launch_device(dxo_ctx* ctx, const dxo_mc_params* prm, int64_t n, const double* deps, const double* sigma_n,
                         double* C_tang, double* sigma, int32_t* niter, double* yielding, double* norm_res, double* dlambda,
                         hipStream_t s);
int dxo_icnn_launch_device(dxo_ctx* ctx, const dxo_icnn* m, int precision, int64_t n, const double* F, double* dP, double* P,
                           hipStream_t s);
int dxo_isihara_launch_device(dxo_ctx* ctx, const dxo_isihara_params* prm, int64_t n, const double* F, double* dP, double* P,
                              hipStream_t s);
// vm_field.hip: fused strain + return map + element vectors of the returned stress (fe[node][cell][i]) — dxo_von_mises_residual
bool dxo_vmf_residual_eligible(const dxo_mesh* mesh);
int dxo_vmf_residual_launch(dxo_ctx* ctx, const dxo_vm_params* prm, const dxo_mesh* mesh, const double* u, const double* sigma_n,
                            const double* p, double* sigma, double* dp, double* fe, hipStream_t s);

// Device mirror of the Mohr-Coulomb history variable plus the stress of the last call (include/dxo.h, dxo_mc_state_*):
// ONE allocation, [sigma_n n*4 | sigma n*4] doubles.
struct dxo_mc_state {
    int64_t n = 0;
    void* blob = nullptr;
    double *sigma_n = nullptr, *sigma = nullptr;
    bool uploaded = false, has_result = false;
};
