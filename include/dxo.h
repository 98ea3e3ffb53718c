/*
 * dxo.h — C ABI of libdxo_hip.so: MI355X (gfx950) quadrature-point kernels that stand
 * behind dolfinx-external-operator's `external_function(derivatives)(*operand_arrays)`
 * callback (reference: src/dolfinx_external_operator/external_operator.py:432).
 *
 * Every entry point is plain C: pointers, sizes, POD parameter structs. No torch / numpy /
 * DOLFINx types cross this boundary. All arrays are C-contiguous, cell-major ->
 * quadrature-point -> component ("AoS per point"), exactly the layout the reference hands to
 * and expects back from the user kernel (external_operator.py:286-290, 393-402).
 *
 * Conventions
 *   - every function returns 0 on success, <0 for an invalid argument (DXO_E_*), >0 for a
 *     HIP runtime error code (hipError_t). dxo_last_error(ctx) gives the text.
 *   - `mem` says where ALL data pointers of that call live: DXO_MEM_HOST (pageable or pinned
 *     host memory: the library does H2D -> kernel -> D2H, chunked and overlapped) or
 *     DXO_MEM_DEVICE (device memory on the ctx's GPU: the kernel is launched on the ctx
 *     stream and the call returns without synchronising).
 *   - the library never frees or keeps caller memory. Scratch is owned by the ctx.
 *   - non-convergence of a local Newton solve is NOT an error (the reference only reports
 *     niter / norm_res, demo_plasticity_mohr_coulomb.py:584-591).
 *   - one ctx per (process, device). Every entry point takes the ctx's mutex, so calls on one ctx
 *     from several host threads are serialised by the library (the reference calls from a single
 *     Python thread inside the SNES callback, petsc/petsc.py:60); use one ctx per thread for
 *     concurrency.
 */
#ifndef DXO_H
#define DXO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DXO_ABI_VERSION 2

/* error codes (negative = caller error) */
#define DXO_OK 0
#define DXO_E_NULL (-1)      /* required pointer is NULL            */
#define DXO_E_DIM (-2)       /* unsupported tensor dimension d/gdim */
#define DXO_E_SIZE (-3)      /* negative size                       */
#define DXO_E_MEM (-4)       /* bad `mem` selector                  */
#define DXO_E_ALIGN (-5)     /* device pointer not 8-byte aligned   */
#define DXO_E_OPTION (-6)    /* unknown option / bad value          */
#define DXO_E_NODEVICE (-7)  /* no usable HIP device                */

#define DXO_MEM_HOST 0
#define DXO_MEM_DEVICE 1

typedef struct dxo_ctx dxo_ctx;

/* Material constants of the von Mises demo (demo_plasticity_von_mises.py:185-188);
 * lmbda, mu, C_elas, deviatoric (:190-204) are derived from them inside the kernel. */
typedef struct dxo_vm_params {
    double E;        /* Young modulus            (70e3)               */
    double nu;       /* Poisson ratio            (0.3)                */
    double sigma_0;  /* yield strength           (250)                */
    double H;        /* hardening modulus        (E*Et/(E-Et), Et=E/100) */
} dxo_vm_params;

/* Constants of the Mohr-Coulomb demo (demo_plasticity_mohr_coulomb.py:110-116, 469). */
typedef struct dxo_mc_params {
    double E;        /* 6778                                          */
    double nu;       /* 0.25                                          */
    double c;        /* cohesion 3.45                                 */
    double phi;      /* friction angle  [rad]                         */
    double psi;      /* dilatancy angle [rad]                         */
    double theta_T;  /* Abbo-Sloan transition angle [rad]             */
    double a;        /* tension cut-off parameter 0.26 c / tan(phi)   */
    double tol;      /* relative residual tolerance 1e-8 (:469)       */
    int32_t nitermax;/* 200 (:469)                                    */
    int32_t _pad;
} dxo_mc_params;

/* Timing of the last call on a ctx, milliseconds (HIP events on the ctx streams). */
typedef struct dxo_timing {
    double h2d_ms;     /* DXO_MEM_HOST only, else 0                   */
    double kernel_ms;  /* sum over chunks                             */
    double d2h_ms;     /* DXO_MEM_HOST only, else 0                   */
    double total_ms;   /* first enqueue -> last completion            */
} dxo_timing;

typedef struct dxo_device_info {
    char name[128];
    char arch[32];          /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    int32_t compute_units;
    int32_t wavefront_size;
    int64_t total_mem_bytes;
} dxo_device_info;

/* ---- context ------------------------------------------------------------------------- */
int dxo_abi_version(void);
int dxo_device_count(int* count);
int dxo_ctx_create(int device, dxo_ctx** out);
int dxo_ctx_destroy(dxo_ctx* ctx);
const char* dxo_last_error(const dxo_ctx* ctx);
int dxo_ctx_device_info(dxo_ctx* ctx, dxo_device_info* info);
/* Borrow a caller-owned hipStream_t for DXO_MEM_DEVICE launches (NULL = library stream). */
int dxo_ctx_set_stream(dxo_ctx* ctx, void* hip_stream);
int dxo_ctx_synchronize(dxo_ctx* ctx);
/* Integer tuning knobs: "vm_variant" (0 scalar AoS kernel, 1 LDS-staged coalesced kernel,
 * default 1), "host_chunk_points" (points per H2D/kernel/D2H pipeline chunk),
 * "nontemporal" (0/1 streaming stores), "timing" (0/1 record dxo_timing on device calls),
 * "blocks_per_cu" (0 = one tile per wave — except that dxo_von_mises / dxo_vm_expand_tangent writing into a block from
 * dxo_vm_output_alloc use the launch shape its calibration found best; k = grid-stride over k workgroups per CU),
 * "mc_variant" (0 lane-per-point Newton; 1 classify kernel + compacted Newton kernel with lane refill; 2 — the default —
 * classification and Newton in one persistent kernel, plastic points queued per wave in LDS; outputs bit-identical),
 * "mc_blocks_per_cu" (variant 1: persistent Newton workgroups per CU, default 3), "mc_waves_per_simd" (variant 1),
 * "mc_part_points" (variant 1: points per classify/Newton pass, default and maximum 2^30: the compacted list holds int32
 * entries; variant 2 splits batches beyond 2^30 points by itself), "icnn_variant"
 * (fp32 network: 0 lane-per-point VALU kernel; 1 MFMA kernel on fp32-input MFMA; 2 — the default — the same GEMMs with
 * every fp32 operand split exactly into three bf16 numbers and six partial products on the bf16 MFMA pipe: fp32-level
 * results (it agrees with the oracle as closely as variant 1) in about 0.68 of the time; 3 and 4 — two kernels measured
 * slower in round 5 — exist only in a -DDXO_EXPERIMENTS build, scripts/exp/icnn_variants.h: this library answers them, like
 * "adjoint_patch" = 1, with DXO_E_OPTION), "host_small_bytes" (host batches whose inputs + outputs
 * fit this many bytes — default 8 MiB — and are not split by "host_chunk_points" take the SMALL path instead of the chunked
 * pipeline: the fixed cost per call at the reference's demo sizes; 0 switches the path off; per-phase dxo_timing is recorded
 * on it only with "timing" = 1, through one packed H2D and one packed D2H copy), "host_zero_copy_bytes" (default 8 MiB: on
 * that path, batches up to this size are not copied by DMA at all — the kernel reads and writes page-locked, device-mapped
 * HOST memory over PCIe directly: a caller's array that lies in a page-locked block this library handed out or registered
 * (dxo_host_alloc, dxo_host_register) is used IN PLACE, any other array goes through the context's pinned staging block (one
 * host memcpy each way); arrays that live on the device (dxo_vm_state) are used where they are; the batch runs in up to four
 * pieces of about "host_zero_copy_piece_bytes" (default 1 MiB, 0 = one piece), piece k + 1 being packed while piece k's kernel
 * runs. Results are bit-identical; 0 = off; with "timing" = 1 the copy form is used so that the phases can be timed; only the
 * operators whose kernels read every input once and store whole lines take this form — von Mises, heat, conductivity, Isihara
 * and their fused-operand forms; the Mohr-Coulomb and network kernels re-read inputs / store partial lines and keep the two
 * packed DMA copies. Measured (profiles/r06_demo_latency.txt): the von Mises demo's 15 000 points 194 -> 98 us per call),
 * "vm_host_tangent" (DXO_MEM_HOST dxo_von_mises: 0 = C_tang comes
 * back over PCIe, bit-identical to a device call; 1 = only (sigma, dp) cross PCIe and the caller's C_tang array is
 * rebuilt from them by the context's host threads while later chunks are in flight — same formulas as
 * dxo_vm_expand_tangent, agrees with the device tangent to rounding (5e-16 of its scale measured); the reference's 0/0
 *  point at f_el == 0 exactly is carried across in the sign bit of dp and comes out as NaN like in the copy mode),
 * "vm_mark_indeterminate" (default 0; 1 = DXO_MEM_DEVICE von Mises calls return dp = -0.0 — numerically 0 — at a point
 * with f_elastic == 0 EXACTLY, where the reference's n_elas is 0/0 and its tangent all NaN, demo_plasticity_von_mises.py:318,
 * so that dxo_vm_expand_tangent can reproduce that NaN from (sigma, dp); dxo_vm_clear_marks turns the marks back into +0;
 * the compact multi-GPU gathers set it themselves), "host_threads" (worker threads of that host half, default 32, capped by the CPUs the process may use: affinity mask and
 * cgroup CPU quota, minus two; DXO_HOST_CPU_BUDGET in the environment overrides the detection), "vm_rebuild_chunk_points" (pipeline chunk of that mode, default 2^17), "vm_rebuild_min_points" (default 2^16: smaller batches take the copy mode — back to back the rebuild wins from 15 000 points on, but calls that are milliseconds apart pay the wake-up of the host threads), and the "placement_*" options
 * of the output arena below. */
int dxo_ctx_set_option(dxo_ctx* ctx, const char* key, int64_t value);
int dxo_ctx_get_option(dxo_ctx* ctx, const char* key, int64_t* value);
int dxo_last_timing(dxo_ctx* ctx, dxo_timing* t);
/* Pinned host buffers (hipHostMalloc) so DXO_MEM_HOST calls DMA without staging. ctx may be NULL for both
 * (the memory belongs to the process; a buffer may outlive the context it was first used with). */
int dxo_host_alloc(dxo_ctx* ctx, int64_t bytes, void** ptr);
/* Page-lock / release memory the caller owns (hipHostRegister): e.g. the storage of the coefficient the results are
 * written into at every call (x.array of a fem.Function lives as long as the simulation). ctx may be NULL. */
int dxo_host_register(dxo_ctx* ctx, void* ptr, int64_t bytes);
int dxo_host_unregister(dxo_ctx* ctx, void* ptr);
int dxo_host_free(dxo_ctx* ctx, void* ptr);

/* ---- output arena: device memory for OUTPUT arrays, placed by calibration -------------------------------------
 * The kernels are HBM-bound and mostly stores (von Mises d = 6: 344 of 448 B per point). On MI355X the rate of a
 * multi-GB streaming-write sweep depends on the allocation it goes to (pure stores about 5.6-6.0, 6.4-6.5 or 6.9-7.1
 * TB/s; the class is stable for the life of the allocation and belongs to its virtual range, DESIGN.md 3.1,
 * profiles/r02_place_exp*.txt). dxo_output_alloc makes several ordinary allocations side by side, times one
 * streaming-write sweep on each, keeps the first above option "placement_good_GBps" (6800), else the fastest, and
 * frees the rest; the block is owned by the ctx (freed by dxo_output_free or dxo_ctx_destroy). Meant for the
 * persistent coefficient buffers a solver allocates once. Options: "placement_mode" 2 = hipMalloc candidates
 * (default), 0 = plain hipMalloc; "placement_candidates"
 * (default 16, at most 32, and never more than fit 60 % of the free memory together); "placement_min_bytes"
 * (default 1 GiB: smaller blocks are plain hipMalloc, a working set that small lives in the caches); "placement_vmm"
 * (default 1: three of every four candidates — 2: all of them — are virtual ranges backed by 2 MB physical chunks, hipMemCreate / hipMemMap,
 * which land in the fast class about twice as often as hipMalloc blocks and were the only fast ones for vm_tile on several boxes; 0 = hipMalloc candidates only, e.g. for buffers
 * that must be shareable through hipIpcGetMemHandle); "placement_probe" (default 1: candidates are ranked with a sweep in
 * the constitutive kernels' own pattern — three input streams read, three output streams written, 13 : 43 KiB per
 * 64-point tile, rates in algorithmic GB/s, early exit at "placement_good_mix_GBps" = 6250; 0: one stream of stores,
 * early exit at "placement_good_GBps"). A search can come up without a fast block: the winner is REJECTED — and one more search
 * with fresh candidates made while it stays allocated, the two winners then timed head to head — when it is below
 * "placement_accept_pct" (97) per cent of the best rate a calibration of this context ever kept for the same probe and
 * block size, or, without such a record, when it does not stand out from its own candidates (below
 * "placement_standout_pct" = 106 per cent of their median; once; not if it already runs at "placement_good_mix_GBps"); "placement_rounds" (3)
 * bounds the searches; the rate tested is the winner's rate AFTER the other candidates have been freed. The
 * calibration WRITES the block (zeros) and is synchronous. dxo_output_info reports what the calibration saw.
 * dxo_output_free RETAINS one calibrated block (the fastest it has been handed) instead of releasing it, and the next request of
 * exactly its size and probe is given that block back after a re-timing (rounds = 0 in its record: ~10 ms instead of a 2-7 s
 * search, and the only way to a fast block on a box where one range in 68 is fast); it is released when a search starts, when
 * dxo_device_alloc would otherwise fail, and with the context. Option "placement_cache" = 0 frees at once. */
#define DXO_PLACEMENT_MAX 32
typedef struct dxo_placement_info {
    int16_t mode;                           /* how the block was obtained: 0 plain hipMalloc, 2 candidates         */
    int16_t probe_kind;                     /* what the candidates were timed with: 0 one store stream, 1 the
                                               kernels' six-stream read + write sweep (option "placement_probe"),
                                               2 the von Mises kernel itself (dxo_vm_output_alloc), 3 the caller's
                                               launch (dxo_output_alloc_probed)                                     */
    int32_t candidates;                     /* ranges / allocations timed                                          */
    int32_t chosen;                         /* index of the one kept (-1: no calibration)                          */
    uint32_t vmm_mask;                      /* bit k: candidate k was built from 2 MB physical chunks (option
                                               "placement_vmm"), not by hipMalloc                                  */
    double probe_GBps[DXO_PLACEMENT_MAX];   /* streaming-write rate of each candidate                              */
    double calibration_ms;                  /* wall time of the whole call                                         */
    double chosen_GBps;                     /* rate of the block kept, re-timed after all but the three best
                                               candidates were freed (rates read low while many coexist)           */
    int32_t tuned_blocks_per_cu;            /* dxo_vm_output_alloc: launch shape of vm_tile that was fastest on the block
                                               kept (0 one tile per wave, k persistent workgroups per CU); the kernel
                                               uses it whenever it writes into this block and option "blocks_per_cu"
                                               is 0                                                                 */
    int32_t rounds;                         /* candidate searches this calibration made (1-3): a winner below 0.97 of the best
                                               rate the context ever kept for this probe and size, or one that does not stand
                                               out from its own candidates, buys one more search with fresh candidates      */
} dxo_placement_info;
int dxo_output_alloc(dxo_ctx* ctx, int64_t bytes, void** ptr);
int dxo_output_free(dxo_ctx* ctx, void* ptr);
int dxo_output_info(dxo_ctx* ctx, const void* ptr, dxo_placement_info* info);
/* The caller's own consumer as the probe. launch(block, shape, user) must enqueue ONE pass of the kernel that will write
 * the block — typically the dxo_* entry point itself with device pointers into `block` — on the context's launch stream
 * (dxo_ctx_set_stream, else the library's) and return without synchronising; it runs on the calling thread with the
 * context's (recursive) lock held. shapes: launch shapes to try (NULL / 0: one pass with shape 0); the best one is
 * recorded in tuned_blocks_per_cu for the caller to read back. bytes_per_launch: what a pass moves (for the GB/s of the
 * record; 0: the block size). probe_kind 3. */
typedef void (*dxo_probe_launch)(void* block, int shape, void* user);
int dxo_output_alloc_probed(dxo_ctx* ctx, int64_t bytes, dxo_probe_launch launch, void* user, double bytes_per_launch,
                            const int32_t* shapes, int n_shapes, void** ptr);
/* The output arrays of dxo_von_mises for n points of Mandel length d, as ONE arena block calibrated WITH THE KERNEL
 * ITSELF: every candidate is timed running vm_tile on synthetic inputs of the reference's distribution, in two launch
 * shapes (one tile per wave, 32 persistent workgroups per CU), and the block that makes the kernel fastest is kept
 * together with its shape (probe_kind 2). Generic sweeps rank blocks for ONE access pattern: blocks that topped a
 * store-stream or six-stream ranking ran the kernel anywhere between 5.3 and 6.4 TB/s (scripts/exp/archive/arena_eval.hip).
 * C_tang is the base of the block (pass it to dxo_output_free / dxo_output_info); sigma and dp start on 256-byte borders
 * behind it. Same options as dxo_output_alloc; blocks below "placement_min_bytes" are plain hipMalloc. */
int dxo_vm_output_alloc(dxo_ctx* ctx, int d, int64_t n, double** C_tang, double** sigma, double** dp);

/* ---- von Mises radial return + consistent tangent ---------------------------------------
 * Replaces return_mapping/_kernel + C_tang_impl, demo_plasticity_von_mises.py:298-352.
 *   d        Mandel vector length: 4 (plane strain, the reference demo) or 6 (3-D)
 *   n        number of quadrature points (num_cells * nq)
 *   deps     [n][d]   strain increment operand            (in)
 *   sigma_n  [n][d]   stress at the previous load step    (in; closure state, :347)
 *   p        [n]      cumulative plastic strain           (in; closure state, :348)
 *   C_tang   [n][d][d] consistent tangent                 (out; with DXO_MEM_DEVICE it may be NULL: only (sigma, dp)
 *                                                          are written, 160 instead of 448 B/point at d = 6, for callers
 *                                                          that rebuild the tangent with dxo_vm_expand_tangent)
 *   sigma    [n][d]   new stress                          (out)
 *   dp       [n]      plastic strain increment            (out)
 */
int dxo_von_mises(dxo_ctx* ctx, const dxo_vm_params* prm, int d, int64_t n, int mem,
                  const double* deps, const double* sigma_n, const double* p,
                  double* C_tang, double* sigma, double* dp);

/* Consistent tangent rebuilt from the RETURNED state: C_tang[n][d][d] from sigma[n][d], dp[n] (same formulas,
 * demo_plasticity_von_mises.py:318-324, with s = dev sigma). The multi-GPU gather exchanges (sigma, dp) and rebuilds
 * every tangent with this entry point. Agrees with dxo_von_mises' C_tang to rounding (<= 1e-14 of its scale; elastic points
 * bit for bit). A point whose dp is -0.0 (the producer's mark for the reference's 0/0 at f_elastic == 0, option
 * "vm_mark_indeterminate") comes out all NaN like the reference's tangent; without the mark such a point is C_elas. */
int dxo_vm_expand_tangent(dxo_ctx* ctx, const dxo_vm_params* prm, int d, int64_t n, int mem,
                          const double* sigma, const double* dp, double* C_tang);
/* dp[n] in DEVICE memory: every -0.0 (mark, see above) becomes +0.0, the value the reference holds there. Asynchronous on
 * the ctx stream. */
int dxo_vm_clear_marks(dxo_ctx* ctx, int64_t n, double* dp);

/* History update at the end of a load step, DEVICE memory only (demo_plasticity_von_mises.py:564-565):
 *   p[n] += dp[n];  sigma_n[n][d] = sigma[n][d].   One fused pass, asynchronous on the ctx stream. */
int dxo_vm_commit_state(dxo_ctx* ctx, int d, int64_t n, double* p, const double* dp,
                        double* sigma_n, const double* sigma);

/* ---- History variables resident on the device -------------------------------------------------
 * The reference's callback re-reads sigma_n and p from closure-captured host arrays at every call
 * (demo_plasticity_von_mises.py:347-348) although they change only at the end of a load step (:564-565). A dxo_vm_state
 * is their device mirror for n points of Mandel length d, plus the (sigma, dp) of the last call:
 *   dxo_vm_state_upload    host (or device) arrays -> mirror; blocking. Call it once, and again whenever the caller
 *                          changed its arrays in any other way than the load-step update below.
 *   dxo_von_mises_state    dxo_von_mises with sigma_n, p read from the mirror. `mem` says where deps and the outputs
 *                          live. DXO_MEM_HOST: only deps goes up (48 of 104 B/point at d = 6); option vm_host_tangent
 *                          applies as in dxo_von_mises. DXO_MEM_DEVICE: sigma / dp may be NULL (results stay in the
 *                          mirror, see dxo_vm_state_pointers).
 *   dxo_vm_state_commit    p += dp; sigma_n <- sigma inside the mirror with the results of the LAST call — what the
 *                          caller does to its host arrays at :564-565. Blocking. DXO_E_SIZE if there was no call since
 *                          the last upload / commit.
 *   dxo_vm_state_download  mirror -> caller's arrays (either may be NULL): checks and checkpoints.
 *   dxo_vm_state_pointers  the device arrays themselves: sigma_n[n][d], p[n], sigma[n][d], dp[n] (any may be NULL).
 * Results are bit-identical to dxo_von_mises on the same values. One state belongs to one ctx. */
typedef struct dxo_vm_state dxo_vm_state;
int dxo_vm_state_create(dxo_ctx* ctx, int d, int64_t n, dxo_vm_state** out);
void dxo_vm_state_destroy(dxo_ctx* ctx, dxo_vm_state* state);
int dxo_vm_state_upload(dxo_ctx* ctx, dxo_vm_state* state, int mem, const double* sigma_n, const double* p);
int dxo_vm_state_download(dxo_ctx* ctx, dxo_vm_state* state, int mem, double* sigma_n, double* p);
int dxo_vm_state_commit(dxo_ctx* ctx, dxo_vm_state* state);
int dxo_vm_state_pointers(dxo_ctx* ctx, dxo_vm_state* state, double** sigma_n, double** p, double** sigma, double** dp);
int dxo_von_mises_state(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_vm_state* state, int mem, const double* deps,
                        double* C_tang, double* sigma, double* dp);

/* Device memory owned by the caller but allocated through the library (hipMalloc / hipFree / hipMemcpyAsync on
 * the ctx stream + synchronise), so a host program without its own HIP binding can keep state on the GPU.
 * kind: 0 host->device, 1 device->host, 2 device->device. */
int dxo_device_alloc(dxo_ctx* ctx, int64_t bytes, void** ptr);
int dxo_device_free(dxo_ctx* ctx, void* ptr);
int dxo_copy(dxo_ctx* ctx, void* dst, const void* src, int64_t bytes, int kind);

/* ---- nonlinear heat flux ------------------------------------------------------------------
 * Replaces k / q_impl / dqdT_impl / dqdsigma_impl, demo_nonlinear_heat_equation_part2.py:209-261
 * (same code: test/test_external_operators_evaluation.py:64-86).
 *   T [n], sigma [n][gdim] (in);  q [n][gdim], dqdT [n][gdim], dqdsigma [n][gdim][gdim] (out).
 * Any output may be NULL (the reference evaluates the three derivatives as separate
 * operators; one fused launch fills whichever are requested). gdim in {1,2,3}.
 */
int dxo_heat(dxo_ctx* ctx, double A, double B, int gdim, int64_t n, int mem,
             const double* T, const double* sigma,
             double* q, double* dqdT, double* dqdsigma);

/* Scalar conductivity of the part-1 heat demo: k[n] = 1 / (A + B T[n]) and dkdT[n] = -B k^2 (k_impl / dkdT_impl,
 * demo_nonlinear_heat_equation_part1.py:251-271). Either output may be NULL; one fused pass. The operator lives on a CG
 * space in the demo: its values reach the coefficient through the dofmap assigner (dxo_assign below). */
int dxo_conductivity(dxo_ctx* ctx, double A, double B, int64_t n, int mem, const double* T, double* k, double* dkdT);

/* ---- Mohr-Coulomb (Abbo-Sloan) return mapping + AD-through-Newton tangent ----------------------
 * Replaces return_mapping / dsigma_ddeps_vec / C_tang_impl,
 * demo_plasticity_mohr_coulomb.py:282-391, 405-462, 469-533, 555, 574-593.
 *   deps [n][4], sigma_n [n][4] (in; sigma_n is closure state, :579);
 *   C_tang [n][4][4] = d sigma / d deps differentiated THROUGH the Newton iterations (jacfwd of the
 *   while_loop, :555), sigma [n][4] (out).
 *   optional diagnostics (NULL to skip) = the reference's aux outputs (:533, :582):
 *   niter [n] int32, yielding [n] = f(sigma_n + C deps), norm_res [n], dlambda [n].
 * A point that does not converge within prm->nitermax iterations is not an error: niter == nitermax
 * and norm_res tell (the reference prints the same, :584-591).
 * DXO_MEM_DEVICE needs 16-byte aligned deps/sigma_n/C_tang/sigma. */
int dxo_mohr_coulomb(dxo_ctx* ctx, const dxo_mc_params* prm, int64_t n, int mem,
                     const double* deps, const double* sigma_n,
                     double* C_tang, double* sigma,
                     int32_t* niter, double* yielding, double* norm_res, double* dlambda);

/* History variable resident on the device, the Mohr-Coulomb counterpart of dxo_vm_state: the
 * reference's callback re-reads sigma_n from a closure-captured host array at every call
 * (demo_plasticity_mohr_coulomb.py:579) although it changes only at the end of a load step
 * (:728, `sigma_n.x.array[:] = sigma.ref_coefficient.x.array`). A dxo_mc_state mirrors sigma_n[n][4] and keeps the
 * stress of the last call:
 *   dxo_mc_state_upload     caller's array -> mirror; blocking. Once, and again after any change other than the commit.
 *   dxo_mohr_coulomb_state  dxo_mohr_coulomb with sigma_n read from the mirror; `mem` says where deps and the outputs live
 *                           (DXO_MEM_HOST: only deps goes up, 32 of 64 B/point; DXO_MEM_DEVICE: sigma may be NULL, the
 *                           result stays in the mirror, see dxo_mc_state_pointers).
 *   dxo_mc_state_commit     sigma_n <- sigma of the LAST call inside the mirror (:728). Blocking. DXO_E_SIZE if there was
 *                           no call since the last upload / commit; a state of n = 0 points commits trivially.
 *   dxo_mc_state_download   mirror -> caller's array (checks, checkpoints);  dxo_mc_state_pointers: the device arrays.
 * Results are bit-identical to dxo_mohr_coulomb on the same values. One state belongs to one ctx. */
typedef struct dxo_mc_state dxo_mc_state;
int dxo_mc_state_create(dxo_ctx* ctx, int64_t n, dxo_mc_state** out);
void dxo_mc_state_destroy(dxo_ctx* ctx, dxo_mc_state* state);
int dxo_mc_state_upload(dxo_ctx* ctx, dxo_mc_state* state, int mem, const double* sigma_n);
int dxo_mc_state_download(dxo_ctx* ctx, dxo_mc_state* state, int mem, double* sigma_n);
int dxo_mc_state_commit(dxo_ctx* ctx, dxo_mc_state* state);
int dxo_mc_state_pointers(dxo_ctx* ctx, dxo_mc_state* state, double** sigma_n, double** sigma);
int dxo_mohr_coulomb_state(dxo_ctx* ctx, const dxo_mc_params* prm, dxo_mc_state* state, int mem, const double* deps,
                           double* C_tang, double* sigma, int32_t* niter, double* yielding, double* norm_res,
                           double* dlambda);

/* Inner-Newton summary of a dxo_mohr_coulomb call whose diagnostics live in DEVICE memory: the numbers the
 * reference prints at every call (unique iteration counts and their multiplicities, max f, max residual,
 * demo_plasticity_mohr_coulomb.py:584-591), reduced on the GPU (wave __shfl_xor maxima, LDS histogram).
 *   niter [n] int32 (device); yielding / norm_res [n] (device, may be NULL)
 *   hist [nbins] (host, out): hist[k] = #points with niter == k (k >= nbins-1 clamps into the last bin);
 *   nbins <= 1024, use prm->nitermax + 1.   max_* (host, out, may be NULL): maxima over non-NaN entries,
 *   -inf if none; nan_counts[2] (host, out, may be NULL): NaN entries of yielding / norm_res. Blocking. */
int dxo_mc_summary(dxo_ctx* ctx, int64_t n, const int32_t* niter, const double* yielding, const double* norm_res,
                   int nbins, int64_t* hist, double* max_yielding, double* max_norm_res, int64_t* nan_counts);

/* ---- ICNN hyperelastic surrogate: stress P = dW/dF and tangent dP/dF --------------------------
 * Replaces ICNN.forward + compute_stress_local + vmap(jacfwd(.)) + dP_dF_impl,
 * demo_hyperelasticity.py:256-300, 362-381, 429-456. The weight struct takes the tensors of the
 * reference's state_dict as they are stored (fp32, row-major [out][in]); softplus on the convex
 * layers (:238) and the fold of the activation-free layer 0 into layer 1 happen inside create().
 * Architecture is the demo's: n_input 3, n_hidden [64, 64, 64], n_output 1 (:302-307).
 *   F  [n][4] = (F11, F12, F21, F22) (in);  dP [n][4][4] with dP[i][j] = dP_i/dF_j, P [n][4] (out),
 *   the reference's return order is (dP, P) (:456).
 *   precision 0: network in fp32 like the reference (`.float()`, :286); 1: network in fp64
 *   (BASELINE config 5's fp64-vs-fp32 tolerance study). Features and chain rule are fp64 in both. */
typedef struct dxo_icnn_weights {
    const float* layers0_weight;   /* layers.0.weight        [64][3]  */
    const float* layers0_bias;     /* layers.0.bias          [64]     */
    const float* layers1_weights;  /* layers.1.weights       [64][64] */
    const float* skip1_weight;     /* skip_layers.1.weight   [64][3]  */
    const float* skip1_bias;       /* skip_layers.1.bias     [64]     */
    const float* layers2_weights;  /* layers.2.weights       [64][64] */
    const float* skip2_weight;     /* skip_layers.2.weight   [64][3]  */
    const float* skip2_bias;       /* skip_layers.2.bias     [64]     */
    const float* layers3_weights;  /* layers.3.weights       [1][64]  */
    const float* skip3_weights;    /* skip_layers.3.weights  [1][3]   */
    int32_t n_hidden;              /* 64                               */
    int32_t _pad;
} dxo_icnn_weights;
typedef struct dxo_icnn dxo_icnn;
int dxo_icnn_create(dxo_ctx* ctx, const dxo_icnn_weights* w, dxo_icnn** out);
int dxo_icnn_destroy(dxo_ctx* ctx, dxo_icnn* model);
/* H_flat[4] of the stress correction P_cor = F @ H (:371), computed by create() at F = I. */
int dxo_icnn_correction(dxo_ctx* ctx, const dxo_icnn* model, double* H_flat);
int dxo_icnn_eval(dxo_ctx* ctx, const dxo_icnn* model, int precision, int64_t n, int mem,
                  const double* F, double* dP, double* P);

/* Analytic Isihara energy the network was trained on (demo_hyperelasticity.py:686-703; UFL-only in the reference):
 *   W = c1 (I1bar-3) + c2 (I2bar-3) + c3 (I1bar-3)^2 + c4 (J-1)^2,  reference constants (0.5, 1.0, 1.0, 1.5).
 * Same I/O as dxo_icnn_eval: F[n][4] -> dP[n][4][4] (dP_i/dF_j), P[n][4], fp64 throughout. det F <= 0 -> NaN. */
typedef struct dxo_isihara_params { double c1, c2, c3, c4; } dxo_isihara_params;
int dxo_isihara(dxo_ctx* ctx, const dxo_isihara_params* prm, int64_t n, int mem,
                const double* F, double* dP, double* P);

/* ---- operand evaluation on the device ----------------------------------------
 * Device counterpart of `fem.Expression(operand, points).eval(mesh, entities)` as evaluate_operands calls it
 * (src/dolfinx_external_operator/external_operator.py:386-402) for operands that are linear in the value or the
 * gradient of ONE Lagrange field (every operand of the reference's demos). The mesh/element description is
 * uploaded once (dxo_mesh_create); each call gathers the field's dofs and writes (n_cells, nq, value_size)
 * doubles in Expression.eval's C order.
 *   dofmap       = V.dofmap.list            [num_cells][ndofs]  node indices (unblocked); field vector u is
 *                                            blocked: u[node*bs + i], as fem.Function.x.array
 *   geom_dofmap  = mesh.geometry.dofmap     [num_cells][ngeom]
 *   x            = mesh.geometry.x          [num_geom_nodes][x_stride]  (DOLFINx: x_stride = 3), first gdim used
 *   phi, dphi    = V.element.basix_element.tabulate(1, points): values [nq][ndofs], reference gradients
 *                  [nq][ndofs][gdim]; dpsi = the coordinate element's reference gradients [nq][ngeom][gdim]
 * Cells are not assumed affine: J is rebuilt at every point from dpsi.
 * Block sizes: value / grad / value_grad act on each component alone and take ANY bs in 1..DXO_OPERAND_MAX_BS (the reference hands
 * `fem.Expression` whatever field the operand is: test/test_nested_ex_op.py:113-118 evaluates a 4-component DG field): bs = 1 and bs = gdim run
 * the dense kernels, any other bs one scalar launch per component. The kinds built from grad u of a displacement-like field need bs = gdim.
 * The adjoint entry points take bs = 1 or gdim. */
#define DXO_OPERAND_MAX_BS 64
enum {
    DXO_OPERAND_VALUE = 0,       /* u                       value_size = bs                         (heat: T)        */
    DXO_OPERAND_GRAD = 1,        /* grad u, [i][j]=du_i/dx_j value_size = bs*gdim                    (heat: grad T)   */
    DXO_OPERAND_EPS_MANDEL = 2,  /* [g00,g11,0,r(g01+g10)] (2-D, demo_plasticity_von_mises.py:225-227),
                                    [g00,g11,g22,r(g01+g10),r(g02+g20),r(g12+g21)] (3-D), r = sqrt(2)/2; bs = gdim   */
    DXO_OPERAND_DEFGRAD = 3,     /* I + grad u, row-major   value_size = gdim*gdim (demo_hyperelasticity.py:479)      */
    DXO_OPERAND_VALUE_GRAD = 4,  /* [u (bs), grad u (bs*gdim)] in one pass, value_size = bs*(1+gdim) (heat: T and grad T)   */
    /* NONLINEAR operands of one vector field (bs = gdim), formed per point from F = I + grad u — the operand of the reference's own
     * operand test, test/test_operands_evaluation.py:32-36 (F = Identity(d) + grad(u); C = F.T * F; J = det(F); I1 = tr(C)).
     * Forward evaluation only (dxo_eval_operand, dxo_eval_operand_facets): dxo_operand_adjoint answers DXO_E_OPTION for them
     * (the adjoint of a nonlinear operand is a linearisation, which UFL derives on the reference side). */
    DXO_OPERAND_CAUCHY_GREEN = 5,/* C = F^T F, row-major     value_size = gdim*gdim                                          */
    DXO_OPERAND_I1 = 6,          /* tr(F^T F) = sum F_ij^2   value_size = 1                                                  */
    DXO_OPERAND_DETF = 7,        /* det F                    value_size = 1                                                  */
    DXO_OPERAND_DIV = 8          /* div u = tr(grad u)       value_size = 1, bs = gdim; LINEAR: forward and adjoint
                                    (the operand of test/test_external_operators_evaluation.py:141)                           */
};
typedef struct dxo_mesh_desc {
    int32_t gdim;               /* 2 or 3 (= topological dimension) */
    int32_t nq;                 /* quadrature points per cell, <= 64 */
    int32_t ndofs;              /* scalar basis functions per cell of the field's element */
    int32_t ngeom;              /* basis functions per cell of the coordinate element */
    int32_t x_stride;           /* doubles per node in x (0 = gdim) */
    int32_t _pad;
    int64_t num_cells, num_field_nodes, num_geom_nodes;
    const double* phi;
    const double* dphi;
    const double* dpsi;
    const int32_t* dofmap;
    const int32_t* geom_dofmap;
    const double* x;
} dxo_mesh_desc;
typedef struct dxo_mesh dxo_mesh;
int dxo_mesh_create(dxo_ctx* ctx, const dxo_mesh_desc* desc, dxo_mesh** out);   /* host pointers; copies everything */
int dxo_mesh_destroy(dxo_ctx* ctx, dxo_mesh* mesh);
/* value_size of an operand, or a negative error code if (gdim, bs, kind) is unsupported */
int dxo_operand_value_size(int gdim, int bs, int kind);
/* u: field vector (num_field_nodes*bs doubles); cells: entity list or NULL for cells [0, n_cells) (n_cells < 0: all);
 * out: n_cells*nq*value_size doubles. mem applies to u, cells and out alike. */
/* The operand `x = ufl.SpatialCoordinate(mesh)` (test/test_nested_ex_op.py:113-118): the physical position of every quadrature point,
 * x(xi_q) = sum_v psi_v(xi_q) X_v with the coordinate element's own basis. psi = its VALUES at the quadrature points, [nq][ngeom] (the mesh
 * description carries the gradients only), given once; out: n_cells*nq*gdim doubles, cells / n_cells / mem as for dxo_eval_operand. */
int dxo_mesh_set_coordinate_values(dxo_ctx* ctx, dxo_mesh* mesh, const double* psi);
int dxo_eval_coordinate(dxo_ctx* ctx, dxo_mesh* mesh, int mem, const int32_t* cells, int64_t n_cells, double* out);
int dxo_eval_operand(dxo_ctx* ctx, dxo_mesh* mesh, int kind, int bs, int mem, const double* u,
                     const int32_t* cells, int64_t n_cells, double* out);

/* Codim-1 entities. evaluate_operands hands its `entities` to Expression.eval unchanged (external_operator.py:340, 402);
 * for an operator on a facet sub-mesh they are (cell, local_facet) pairs (test/test_codim_external_operator.py:76-84, 111)
 * and the points are quadrature points of the reference FACET mapped into the reference cell per local facet.
 * dxo_mesh_set_facet_tables uploads the tables tabulated at those mapped points, one set per local facet:
 *   phi [n_local_facets][nq][ndofs], dphi [n_local_facets][nq][ndofs][gdim], dpsi [n_local_facets][nq][ngeom][gdim]
 * (host pointers). dxo_eval_operand_facets: entities [n_entities][2] int32 = (cell, local facet); out
 * [n_entities][nq][value_size]; the gradient is the full physical gradient at the facet points. mem applies to u,
 * entities and out alike; with DXO_MEM_DEVICE the entity list is not range-checked. */
int dxo_mesh_set_facet_tables(dxo_ctx* ctx, dxo_mesh* mesh, int n_local_facets, int nq, const double* phi,
                              const double* dphi, const double* dpsi);
int dxo_eval_operand_facets(dxo_ctx* ctx, dxo_mesh* mesh, int kind, int bs, int mem, const double* u,
                            const int32_t* entities, int64_t n_entities, double* out);

/* Operand evaluation FUSED in front of the von Mises return map: one launch for the demo's
 * evaluate_operands + evaluate_external_operators pair (demo_plasticity_von_mises.py:445-456) when the operand is
 * eps(u) of a vector Lagrange field on `mesh` (gdim 2 -> d = 4, gdim 3 -> d = 6). u: num_field_nodes*gdim doubles;
 * state and outputs cover ALL cells of the mesh in cell order: n = num_cells*nq points, same layout and meaning as
 * dxo_von_mises. The strain increment is never written to memory. With DXO_MEM_DEVICE C_tang may be NULL: (sigma, dp) only, for a
 * matrix-free solve that forms the tangent's action from them (dxo_tangent_apply_vm below). */
int dxo_von_mises_field(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, int mem, const double* u,
                        const double* sigma_n, const double* p, double* C_tang, double* sigma, double* dp);
/* The same with the history variables in a dxo_vm_state (n = num_cells*nq, d as the mesh gives it): a host call sends the
 * dof vector only. Option vm_host_tangent applies to both entry points (DXO_MEM_HOST). */
int dxo_von_mises_field_state(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, dxo_vm_state* state, int mem,
                              const double* u, double* C_tang, double* sigma, double* dp);

/* ---- the consumer side on the device, DEVICE memory only ---------------------------------
 * In the reference the coefficient is consumed by DOLFINx assembly of inner(sigma, eps(v)) dx and of the Jacobian
 * form `_apply_derivative_tensor` builds (src/dolfinx_external_operator/external_operator.py:463-486;
 * demo_plasticity_von_mises.py:378-391). These two entry points are the ADJOINT of dxo_eval_operand, so a
 * matrix-free Newton-Krylov solver can leave sigma and C_tang in HBM and move only dof vectors:
 *   dxo_operand_adjoint : out[dof] += sum_q w_q |det J_q| B_q^T s_q   for a quadrature field s of the operand's shape
 *                         (kind EPS_MANDEL with s = sigma: the internal force; GRAD with s = q: the heat residual)
 *   dxo_tangent_apply   : out[dof] += sum_q w_q |det J_q| B_q^T C_tang_q B_q v   (eps / Mandel, bs = gdim), K never formed
 * `out` is ACCUMULATED into (zero it first) — unless option "consumer_overwrite" = 1, with which these calls (and
 * dxo_tangent_*_vm, dxo_tangent_diagonal, dxo_von_mises_residual) SET out to the assembled vector: a Krylov matvec then needs no
 * memset and the node sums no read of out. S / C_tang are laid out like the operator outputs, (n_cells, nq, ...).
 * Full-mesh calls write element vectors and add them per node in the fixed order of the transposed dofmap (no atomics,
 * bit-reproducible); entity subsets, or option "adjoint_atomics" = 1, add with fp64 hardware atomics instead
 * (reproducible to rounding only). Option "adjoint_cell" = 0 switches off the lane-per-cell kernel that the internal
 * force (kind EPS_MANDEL) uses on the standard elements. On hexahedra with the 2x2x2 rule (8 or 27 nodes) the element vectors
 * f_a = sum_q dphi_a(xi_q) . T_q of a wave's 8 cells are formed on the fp64 matrix pipe, [32 x 24] x [24 x 24] as 24
 * v_mfma_f64_16x16x4_f64 (option "adjoint_mfma", default 1; 0 = the round-4 DPP reduce-scatter over a cell's 8 lanes; both are
 * bit-reproducible, they differ from each other in the last bits: a different order of the 8-point sum). The state-based forms
 * (dxo_tangent_apply_vm, dxo_tangent_diagonal_vm) do the same on P2 triangles with the 3-point rule and P2 tetrahedra with the
 * 4-point rule (6 / 9 instructions per wave group; 0 = the lane = (cell, node) loop over tensors parked in LDS).
 * C_tang of dxo_tangent_apply / dxo_tangent_diagonal must be 16-byte aligned (the operator kernels' outputs are).
 * dxo_mesh_set_weights: the nq reference quadrature weights (basix.make_quadrature(...)[1]), host pointer. */
int dxo_mesh_set_weights(dxo_ctx* ctx, dxo_mesh* mesh, const double* weights);
int dxo_operand_adjoint(dxo_ctx* ctx, dxo_mesh* mesh, int kind, int bs, const double* S, const int32_t* cells,
                        int64_t n_cells, double* out);
int dxo_tangent_apply(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, const double* v, double* out);
/* The same two operators for the von Mises operator WITHOUT its tangent array (round 4): the consistent tangent is a function of the
 * returned state (dxo_vm_expand_tangent), so K v and diag(K) are formed from (sigma, dp) — 56 instead of 288 bytes per point at
 * d = 6, ~40 flops per point instead of 36 loads and FMAs. A matrix-free Newton-Krylov solve then never needs C_tang to exist:
 * dxo_von_mises_field / _field_state accept C_tang = NULL on the device path (160 instead of 448 bytes per point) and every Krylov
 * matvec reads a fifth of the bytes. sigma [n][d] (16-byte aligned), dp [n]: the operator's outputs for all cells of the mesh in cell
 * order (e.g. the arrays of dxo_vm_state_pointers); prm: the parameters the operator ran with. Agrees with dxo_tangent_apply on the
 * C_tang of the same call to rounding (1e-13 of the scale; a point the producer marked with dp = -0.0 gives the reference's NaN). */
int dxo_tangent_apply_vm(dxo_ctx* ctx, dxo_mesh* mesh, const dxo_vm_params* prm, const double* sigma, const double* dp,
                         const double* v, double* out);
int dxo_tangent_diagonal_vm(dxo_ctx* ctx, dxo_mesh* mesh, const dxo_vm_params* prm, const double* sigma, const double* dp, double* out);
/* The residual evaluation of a von Mises Newton iteration in ONE call (DEVICE pointers; demo_plasticity_von_mises.py:445-456 followed
 * by the assembly of inner(sigma, eps(v)) dx, :266): (sigma, dp) = return map of (eps(u), sigma_n, p) exactly as dxo_von_mises_field
 * returns them with C_tang = NULL, and R[dof] += sum_q w|J| B^T sigma as dxo_operand_adjoint(EPS_MANDEL) adds it: the two launches back to
 * back on the context's stream. Option "vm_residual_fused" = 1 selects, on Q2 hexahedra with the 2x2x2 rule, ONE kernel that scatters the
 * stress from the registers it was returned in (round 5, with the scatter on the matrix pipe: 1.11 against 1.14-1.16 ms per 10^7 points;
 * R then differs from the two launches' in the last bits — another order of the Jacobian's sums — hence off by default). sigma_n, sigma 16-byte aligned. */
int dxo_von_mises_residual(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, const double* u, const double* sigma_n,
                           const double* p, double* sigma, double* dp, double* R);
/* out[dof] += K_(dof,dof): the diagonal of the same operator (Jacobi preconditioner of a matrix-free Krylov solve). */
int dxo_tangent_diagonal(dxo_ctx* ctx, dxo_mesh* mesh, const double* C_tang, double* out);

/* ---- coefficient assigners on the device, DEVICE memory only --------------------------
 * One scatter for the reference's three dofmap assigners (src/dolfinx_external_operator/external_operator.py):
 * _assign_non_mixed :286-287, _assign_mixed_2d :292-311, _assign_mixed_3d :313-335. For cell c, point p < n_pts,
 * component v < val_size:
 *   coeff[flat_dofs[(c*n_pts + p)*val_size + v]] = values[(c*n_points_total + offset + p)*comp_size + v]
 * Non-mixed: offset = 0, n_points_total = n_pts, comp_size = val_size and flat_dofs = the unrolled dofmap (:18-26);
 * mixed: one call per subspace with its offset / n_pts / val_size (:180-190) and the operator's comp_size (:161).
 * Where several entries target the same dof the LAST one (largest source position) wins, as in NumPy's sequential
 * fancy assignment, so the result is the reference's array, not a race. flat_dofs entries must lie in
 * [0, coeff_size): out-of-range entries are skipped on the device and the call returns DXO_E_SIZE (the NumPy
 * assigner raises IndexError, :287); the check costs one stream synchronisation per call, option
 * "assign_validate" = 0 skips it (entries are still never written out of bounds). The pass that finds the last writer
 * keeps one word per coefficient entry: 32-bit while the entry count is below 2^32 - 1, 64-bit beyond (option
 * "assign_owner_bits" = 64 takes the wide words at any size: the same result, for tests). */
typedef struct dxo_assign_desc {
    int64_t n_cells;
    int32_t n_pts, val_size, offset, n_points_total, comp_size;
    int32_t elem_bytes;     /* width of one scalar of `values` / `coeff`: 4 (float32), 8 (float64), 16 (complex128); 0 = 8. The
                               reference's assigners take whatever scalar type the function space has (float32 / float64 /
                               complex128 in test/test_multiaction.py:15-23): values are moved, never computed on. ABI version 2
                               (version 1 had padding here and moved 8-byte elements only) */
} dxo_assign_desc;
int dxo_assign(dxo_ctx* ctx, const dxo_assign_desc* desc, const int32_t* flat_dofs, const void* values,
               void* coeff, int64_t coeff_size);
/* The same assignment as a PLAN (round 4). Which entry wins a shared dof depends on the dofmap alone, and the dofmap of a
 * function space does not change between the calls of a solve (the reference builds its unrolled dofmaps once, in the
 * operator's constructor, external_operator.py:203-209): dxo_assign_plan_create runs the ownership pass once and keeps, per
 * coefficient entry, the position in `values` of the last entry that targets it; dxo_assign_apply is then ONE gather
 * (coeff[d] = values[src[d]] for the targeted entries, the others keep their value) with coalesced stores and no atomics —
 * bit-identical to dxo_assign at a fraction of its cost (0.60 -> 0.083 ms for 3.4*10^7 entries into 10^7 dofs,
 * profiles/r06_assign_sorted.txt). flat_dofs is device memory and is not kept; out-of-range entries make the creation fail with DXO_E_SIZE. */
typedef struct dxo_assign_plan dxo_assign_plan;
int dxo_assign_plan_create(dxo_ctx* ctx, const dxo_assign_desc* desc, const int32_t* flat_dofs, int64_t coeff_size,
                           dxo_assign_plan** out);
void dxo_assign_plan_destroy(dxo_ctx* ctx, dxo_assign_plan* plan);
int dxo_assign_apply(dxo_ctx* ctx, const dxo_assign_plan* plan, const void* values, void* coeff);   /* elements of the width the plan's descriptor named */
/* A plan of 2^20 coefficient entries or more carries the assignment in two orders (round 6): by coefficient entry (sequential stores, gathered
 * loads) and by position in `values` (near-sequential loads, scattered stores). Which is faster depends on how the caller's dofs are numbered
 * against its cells (Q2 hexahedra 108^3: 0.089 / 0.090 ms with one numbering, 0.107 / 0.085 with another, profiles/r06_assign_sorted.txt), so the
 * FIRST dxo_assign_apply of such a plan launches each form twice on the caller's arrays (every launch leaves the same coefficient), times the
 * second launches and keeps the faster form from then on; a call made while its stream is being captured takes the first form and decides
 * later. Option "assign_plan_form" at creation: 0 = as described, 1 = entry order only, 2 = source order (any size). Returns the form in use
 * (0 = not decided yet, 1, 2) and the two timings in ms (0 until measured); pointers may be NULL. */
int dxo_assign_plan_form(const dxo_assign_plan* plan, double* ms_dof_order, double* ms_source_order);

/* Operand evaluation FUSED in front of the heat-flux kernel: T and sigma = grad T of a scalar Lagrange field on
 * `mesh` are formed per quadrature point and fed to q_impl / dqdT_impl / dqdsigma_impl
 * (demo_nonlinear_heat_equation_part2.py:219-261) in the same launch. T_dofs: num_field_nodes doubles; outputs cover
 * all cells: q[n][gdim], dqdT[n][gdim], dqdsigma[n][gdim][gdim], n = num_cells*nq; any output may be NULL. */
int dxo_heat_field(dxo_ctx* ctx, double A, double B, dxo_mesh* mesh, int mem, const double* T_dofs,
                   double* q, double* dqdT, double* dqdsigma);

/* The Newton / network / analytic hyperelastic operators with the operand formed on the device from the dof vector of
 * the displacement field on `mesh` (gdim = 2): the reference's pair evaluate_operands + evaluate_external_operators
 * (demo_plasticity_mohr_coulomb.py:679-688, demo_hyperelasticity.py:548-557) behind one call.
 *   dxo_mohr_coulomb_field   operand eps(Du) in Mandel notation (:163-165), then dxo_mohr_coulomb's kernels; sigma_n and
 *                            the outputs cover all cells, n = num_cells*nq points, diagnostics as in dxo_mohr_coulomb
 *   dxo_icnn_field           operand F = I + grad u (demo_hyperelasticity.py:479), then dxo_icnn_eval's kernel
 *   dxo_isihara_field        the same operand in front of dxo_isihara
 * u: num_field_nodes*2 doubles (blocked, as fem.Function.x.array). A host caller uploads the dof vector instead of the
 * operand array. dxo_isihara_field (HBM-bound) forms F in the registers of the kernel that evaluates the model, F never
 * reaches memory; for the two compute-bound operators (fp64 Newton, fp32 MFMA network) the operand values pass through a
 * staging buffer of the context in HBM between the operand kernel and theirs (csrc/field_ops.hip). Results are
 * bit-identical to dxo_eval_operand followed by the plain entry point. */
int dxo_mohr_coulomb_field(dxo_ctx* ctx, const dxo_mc_params* prm, dxo_mesh* mesh, int mem, const double* u,
                           const double* sigma_n, double* C_tang, double* sigma, int32_t* niter, double* yielding,
                           double* norm_res, double* dlambda);
int dxo_icnn_field(dxo_ctx* ctx, const dxo_icnn* model, int precision, dxo_mesh* mesh, int mem, const double* u,
                   double* dP, double* P);
int dxo_isihara_field(dxo_ctx* ctx, const dxo_isihara_params* prm, dxo_mesh* mesh, int mem, const double* u,
                      double* dP, double* P);

/* ---- multi-GPU: cell-block sharding + RCCL all-gather inside the library (BASELINE north_star) ----
 * The reference splits only by MPI mesh partition and never gathers quadrature data
 * (src/dolfinx_external_operator/external_operator.py:365-371, 445). Here ONE coefficient vector's quadrature points
 * are split into `world` contiguous, equally sized cell blocks; rank r owns points [r*n_per_rank, (r+1)*n_per_rank).
 * Each GPU evaluates its block and writes straight into its slice of FULL-length output arrays
 * (world*n_per_rank points, device memory on that GPU); RCCL's in-place all-gather over xGMI then gives every GPU the
 * whole vector. All calls are asynchronous on the contexts' streams (dxo_mgpu_synchronize waits).
 *   dxo_mgpu_create       one process, n_dev GPUs (devices == NULL: 0..n_dev-1): a dxo_ctx and a communicator per
 *                         device (ncclCommInitAll); the pointer arrays of the calls below have n_dev entries.
 *   dxo_mgpu_create_rank  one process per GPU: rank 0 calls dxo_mgpu_unique_id, the caller broadcasts the
 *                         DXO_MGPU_ID_BYTES bytes (MPI_Bcast, torch.distributed, a file ...), every rank joins with
 *                         its own ctx; pointer arrays have ONE entry. Collective: all ranks must call.
 * RCCL is loaded at the first dxo_mgpu_* call (dlopen librccl.so.1): DXO_E_NODEVICE if it is absent; RCCL errors are
 * returned as 10000 + ncclResult_t with the text in dxo_mgpu_last_error.
 * gather: DXO_GATHER_NONE (outputs are block-length arrays, no exchange), DXO_GATHER_FULL (all-gather of C_tang,
 * sigma, dp: (d*d+d+1) doubles per point per peer), DXO_GATHER_COMPACT (the kernel writes (sigma, dp) only, those are
 * all-gathered, and EVERY block's tangent — the rank's own too — is rebuilt on each GPU with dxo_vm_expand_tangent:
 * 6.1x fewer link bytes at d = 6; all ranks run the same rebuild on the same gathered values, so the replicas are
 * bit-identical across ranks; they agree with DXO_GATHER_FULL's tangents to rounding, see dxo_vm_expand_tangent, and the
 * reference's NaN tangent at f_elastic == 0 is reproduced on every rank through the dp = -0.0 mark, which is cleared
 * before the call returns). n_per_rank must be even with a gather (16-byte aligned blocks); pad the last block as
 * sharding.CellBlockPartition does. Buffers handed to a collective must be hipMalloc memory: the group's contexts have
 * "placement_vmm" = 0 and the collectives return DXO_E_MEM for a pointer inside a chunk-backed arena block. */
#define DXO_MGPU_ID_BYTES 128
#define DXO_GATHER_NONE 0
#define DXO_GATHER_FULL 1
#define DXO_GATHER_COMPACT 2
/* DXO_GATHER_COMPACT with the exchange of (sigma, dp) as ONE group of ncclSend / ncclRecv pairs (every rank's block straight to
 * every peer: on a fully connected xGMI node each block travels on its own link at once) instead of RCCL's
 * all-gather — same arrays afterwards, bit for bit. dxo_mgpu_von_mises only. */
#define DXO_GATHER_COMPACT_DIRECT 3
/* ... and with the block cut into option "mgpu_chunks" (default 4) pieces on 64-point borders: the kernel of piece k + 1 runs
 * while piece k is on the links (the library's own exchange stream per device), and piece k's tangents are rebuilt while later
 * pieces still travel (SURVEY.md 8e iii). Same arrays afterwards, bit for bit. dxo_mgpu_von_mises only. */
#define DXO_GATHER_COMPACT_PIPELINED 4
typedef struct dxo_mgpu dxo_mgpu;
int dxo_mgpu_create(const int* devices, int n_dev, dxo_mgpu** out);
/* Contexts only, no communicator and no RCCL: for dxo_mgpu_von_mises_host, whose data path has no exchange step. The
 * collective entry points return DXO_E_OPTION on such a group. A device may be listed more than once. */
int dxo_mgpu_create_local(const int* devices, int n_dev, dxo_mgpu** out);
int dxo_mgpu_unique_id(void* id128);
int dxo_mgpu_create_rank(dxo_ctx* ctx, const void* id128, int rank, int world, dxo_mgpu** out);
int dxo_mgpu_destroy(dxo_mgpu* g);
int dxo_mgpu_size(const dxo_mgpu* g);                 /* ranks in the communicator                      */
int dxo_mgpu_local_count(const dxo_mgpu* g);          /* devices driven by this process                 */
int dxo_mgpu_rank(const dxo_mgpu* g, int i);          /* global rank of local device i                  */
dxo_ctx* dxo_mgpu_ctx(dxo_mgpu* g, int i);            /* its context (options, dxo_output_alloc, ...)   */
const char* dxo_mgpu_last_error(const dxo_mgpu* g);
int dxo_mgpu_synchronize(dxo_mgpu* g);
/* In-place all-gather of count_per_rank doubles per rank: buf[i] = full-length array on local device i whose own
 * block already sits at offset rank*count_per_rank. */
int dxo_mgpu_all_gather(dxo_mgpu* g, double* const* buf, int64_t count_per_rank);
/* dxo_von_mises on every local device + the exchange. deps/sigma_n/p[i]: block of local device i (n_per_rank points);
 * C_tang/sigma/dp[i]: full-length arrays on that device (block-length with DXO_GATHER_NONE). */
int dxo_mgpu_von_mises(dxo_mgpu* g, const dxo_vm_params* prm, int d, int64_t n_per_rank, int gather,
                       const double* const* deps, const double* const* sigma_n, const double* const* p,
                       double* const* C_tang, double* const* sigma, double* const* dp);

/* The other pointwise operators sharded the same way (every kernel of the path is a pointwise map): each
 * local device evaluates its cell block (n_per_rank points, the single-GPU entry point on device memory) into its slice of
 * the FULL-length outputs, then one in-place all-gather per requested output array. gather: DXO_GATHER_NONE (block-length
 * outputs, no exchange) or DXO_GATHER_FULL; DXO_GATHER_COMPACT is refused (DXO_E_OPTION): only the von Mises tangent is a
 * function of the other outputs. With a gather n_per_rank must be a multiple of 4. A NULL pointer ARRAY (niter, yielding,
 * ..., q, dqdT, dqdsigma) means the output is not requested. models[i] is the dxo_icnn created on dxo_mgpu_ctx(g, i). */
int dxo_mgpu_mohr_coulomb(dxo_mgpu* g, const dxo_mc_params* prm, int64_t n_per_rank, int gather, const double* const* deps,
                          const double* const* sigma_n, double* const* C_tang, double* const* sigma, int32_t* const* niter,
                          double* const* yielding, double* const* norm_res, double* const* dlambda);
int dxo_mgpu_icnn(dxo_mgpu* g, dxo_icnn* const* models, int precision, int64_t n_per_rank, int gather, const double* const* F,
                  double* const* dP, double* const* P);
int dxo_mgpu_isihara(dxo_mgpu* g, const dxo_isihara_params* prm, int64_t n_per_rank, int gather, const double* const* F,
                     double* const* dP, double* const* P);
int dxo_mgpu_heat(dxo_mgpu* g, double A, double B, int gdim, int64_t n_per_rank, int gather, const double* const* T,
                  const double* const* sigma, double* const* q, double* const* dqdT, double* const* dqdsigma);

/* HOST arrays of all n points sharded over the local devices, no collective: contiguous blocks (borders on 64-point
 * tiles), one dxo_von_mises(DXO_MEM_HOST) per device, concurrently, each over its own PCIe link; results land in the
 * caller's arrays. The NumPy path is PCIe-bound, so the links are what scales it. Works on any group (create,
 * create_local; with create_rank it is the single local device). Options are those of the local contexts (dxo_mgpu_ctx). */
int dxo_mgpu_von_mises_host(dxo_mgpu* g, const dxo_vm_params* prm, int d, int64_t n, const double* deps,
                            const double* sigma_n, const double* p, double* C_tang, double* sigma, double* dp);

/* ---- HBM stream probe (measurement aid, device memory only) --------------------------------
 * Moves data with no arithmetic in the read : write mix of a constitutive kernel, lane-linear 16-byte
 * accesses: n_tiles tiles, each 64 lanes x read_chunks 16-byte loads and 64 x write_chunks 16-byte
 * stores. Supported mixes (read, write): (13,43) von Mises d=6, (9,21) von Mises d=4, (8,20)
 * Mohr-Coulomb, (3,8) heat, (1,1) and (4,4) plain copy. src needs n_tiles*read_chunks*1024 bytes, dst
 * n_tiles*write_chunks*1024 bytes. bench.py reports its GB/s beside the 8 TB/s spec peak. */
int dxo_stream_probe(dxo_ctx* ctx, int read_chunks, int write_chunks, int64_t n_tiles,
                     const void* src, void* dst);

#ifdef __cplusplus
}
#endif
#endif /* DXO_H */
