// assign.hip — the reference's coefficient assigners as one device scatter.
//
// Reference (src/dolfinx_external_operator/external_operator.py):
//   _assign_non_mixed          :286-287   coeff.x.array[unrolled_dofmap] = values
//   _assign_mixed_2d           :292-311   per subspace: coeff.x.array[flat_dofs] = values[:, offset:offset+n_pts]
//   _assign_mixed_3d           :313-335   per subspace: chunk = values[:, offset:offset+n_pts, :][:, :, :val_size]
//                                         coeff.x.array[flat_dofs] = chunk.reshape(n_cells, dofs_per_cell)
// All three are: for cell c, point p < n_pts, component v < val_size
//   coeff[flat_dofs[(c*n_pts + p)*val_size + v]] = values[(c*n_points_total + offset + p)*comp_size + v]
// (non-mixed: offset 0, n_points_total = n_pts, comp_size = val_size; 2-D: comp_size = val_size = 1).
// (_assign_non_mixed_contiguous, :289-290, is a plain copy and needs no kernel: the operator writes in place.)
//
// NumPy's fancy assignment is sequential, so where several entries map to the same dof (continuous spaces: dofs
// shared between cells) the LAST one wins. A parallel scatter would leave whichever store lands last; to return the
// same array as the reference, pass 1 records for every dof the largest source position that targets it
// (atomicMax), pass 2 stores only from that position. HBM/atomic-bound index traffic, no arithmetic.
//
// The reference's assigners are dtype-agnostic (`coeff.x.array[dofs] = values` for whatever scalar type the function space has:
// float32 / float64 / complex128 in test/test_multiaction.py:15-23): values are MOVED, never computed on, so the kernels are
// templates on the element WIDTH — 4, 8 or 16 bytes (dxo_assign_desc::elem_bytes), moved as unsigned words of that width.
#include "dxo_common.h"

namespace {

struct AssignDev {
    int64_t n_cells;
    int n_pts, val_size, offset, n_points_total, comp_size;
};

__device__ __forceinline__ int64_t assign_src(const AssignDev& a, int64_t e) {
    const int per_cell = a.n_pts * a.val_size;
    const int64_t c = e / per_cell;
    const int r = (int)(e - c * per_cell);
    const int p = r / a.val_size, v = r - p * a.val_size;
    return (c * a.n_points_total + a.offset + p) * a.comp_size + v;
}

// The error word (the 8 bytes in front of the owner table): number of flat_dofs entries outside [0, coeff_size). Such entries are skipped by
// both passes (the NumPy assigner the kernel mirrors raises IndexError, external_operator.py:287; here the call
// returns DXO_E_SIZE and coeff holds the in-range part).
// The owner words are 32-bit while the entry count fits (W = uint32_t, every reference-sized mesh) and 64-bit beyond: the pass is bound by
// the NUMBER of device-scope atomics (77 G/s measured, scripts/exp/assign_owner_probe.hip: 0.44 against 0.51 ms on 3.4e7 entries; a load-first
// filter, descending order and atomic-free fixed-point rounds are all slower), the narrow word only halves the table.
template <typename W>
__global__ __launch_bounds__(DXO_BLOCK) void assign_owner(AssignDev a, const int32_t* __restrict__ dofs, W* __restrict__ owner,
                                                          unsigned long long* __restrict__ bad, int64_t coeff_size) {
    const int64_t n = a.n_cells * a.n_pts * a.val_size;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t d = dofs[e];
        if (d < 0 || d >= coeff_size) atomicAdd(bad, 1ull);
        else atomicMax(owner + d, (W)(e + 1));
    }
}

struct alignas(16) assign_u128 { unsigned long long lo, hi; };      // complex128: one 16-byte move

template <typename W, typename T>
__global__ __launch_bounds__(DXO_BLOCK) void assign_store(AssignDev a, const int32_t* __restrict__ dofs, const W* __restrict__ owner,
                                                          const T* __restrict__ values, T* __restrict__ coeff,
                                                          int64_t coeff_size) {
    const int64_t n = a.n_cells * a.n_pts * a.val_size;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int64_t d = dofs[e];
        if (d < 0 || d >= coeff_size) continue;
        if (owner[d] == (W)(e + 1)) coeff[d] = values[assign_src(a, e)];
    }
}

// ---- assignment plans. Which entry wins a shared dof depends on the dofmap alone, and a function space's dofmap never changes
// between the calls of a solve: the plan keeps, for every coefficient entry, the position in `values` of the LAST entry that
// targets it (-1: none), so applying it is one gather with coalesced stores, no atomics and no owner table.
template <typename W, typename I>
__global__ __launch_bounds__(DXO_BLOCK) void assign_plan_finish(AssignDev a, const W* __restrict__ owner, I* __restrict__ src, int64_t coeff_size) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < coeff_size; d += stride) {
        const unsigned long long o = owner[d];
        src[d] = o ? (I)assign_src(a, (int64_t)o - 1) : (I)-1;
    }
}

template <typename I, typename T>
__global__ __launch_bounds__(DXO_BLOCK) void assign_apply(const I* __restrict__ src, const T* __restrict__ values,
                                                          T* __restrict__ coeff, int64_t coeff_size) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < coeff_size; d += stride) {
        const I s = src[d];
        if (s >= 0) coeff[d] = values[s];
    }
}

// ---- the plan in SOURCE order (round 6). The dof-order gather above reads one 8-byte value per coefficient entry from wherever the winning
// entry sits in `values`: how many 64-byte lines that drags depends on how the caller's dofs are numbered against its cells (Q2 hexahedra
// 108^3: 287 MB of loads when neighbouring dofs come from neighbouring local nodes, 354 MB when they come from local nodes 9 apart — 0.089
// against 0.107 ms, scripts/exp/assign_sorted_probe.hip). Walking the winners in the order they lie in `values` makes the loads near-sequential
// whatever the numbering (179 MB, 0.085-0.090 ms) and the 8-byte stores the scattered side. Neither form wins everywhere (a numbering without
// locality favours coalesced stores), so a plan carries both and the first dxo_assign_apply times them on the caller's own arrays.
// Winners are compacted in entry order = source order (assign_src is increasing in e): counts per tile, a scan of the tile counts, then the
// same tiles write their pairs. Deterministic: the plan's layout depends on the dofmap alone.
#ifndef DXO_ASSIGN_CAP
#define DXO_ASSIGN_CAP 16      // workgroups per CU of the direct form (owner + store passes)
#endif
constexpr int ASSIGN_TILE = DXO_BLOCK * 8;      // entries per workgroup tile

template <typename W>
__device__ __forceinline__ bool assign_wins(const int32_t* __restrict__ dofs, const W* __restrict__ owner, int64_t e, int64_t n, int64_t coeff_size) {
    if (e >= n) return false;
    const int64_t d = dofs[e];
    return d >= 0 && d < coeff_size && owner[d] == (W)(e + 1);
}

template <typename W>
__global__ __launch_bounds__(DXO_BLOCK) void assign_tile_count(int64_t n, const int32_t* __restrict__ dofs, const W* __restrict__ owner,
                                                               int64_t coeff_size, uint32_t* __restrict__ tile_count) {
    __shared__ uint32_t part[DXO_BLOCK / DXO_WAVE];
    const int64_t base = (int64_t)blockIdx.x * ASSIGN_TILE;
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < ASSIGN_TILE / DXO_BLOCK; ++k) mine += assign_wins(dofs, owner, base + k * DXO_BLOCK + threadIdx.x, n, coeff_size) ? 1u : 0u;
    for (int o = DXO_WAVE / 2; o > 0; o >>= 1) mine += __shfl_down(mine, o, DXO_WAVE);
    if ((threadIdx.x & (DXO_WAVE - 1)) == 0) part[threadIdx.x / DXO_WAVE] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < DXO_BLOCK / DXO_WAVE; ++w) t += part[w];
        tile_count[blockIdx.x] = t;
    }
}

// exclusive scan of the tile counts in place (one workgroup; a tile holds 2 048 entries, so 3.4e7 entries are 16 608 counts), total behind them
__global__ __launch_bounds__(1024) void assign_tile_scan(uint32_t* __restrict__ tile_count, int64_t n_tiles, unsigned long long* __restrict__ total) {
    __shared__ unsigned long long sums[1024];
    const int64_t per = (n_tiles + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    unsigned long long acc = 0;
    for (int64_t i = lo; i < hi; ++i) acc += tile_count[i];
    sums[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) { const unsigned long long v = sums[t]; sums[t] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = sums[threadIdx.x];
    for (int64_t i = lo; i < hi; ++i) { const uint32_t v = tile_count[i]; tile_count[i] = (uint32_t)run; run += v; }
}

template <typename W, typename I>
__global__ __launch_bounds__(DXO_BLOCK) void assign_tile_pairs(AssignDev a, int64_t n, const int32_t* __restrict__ dofs, const W* __restrict__ owner,
                                                               int64_t coeff_size, const uint32_t* __restrict__ tile_start,
                                                               I* __restrict__ src_s, int32_t* __restrict__ dst_s) {
    __shared__ uint32_t part[DXO_BLOCK / DXO_WAVE];
    const int64_t base = (int64_t)blockIdx.x * ASSIGN_TILE;
    const int lane = threadIdx.x & (DXO_WAVE - 1), wave = threadIdx.x / DXO_WAVE;
    uint32_t at = tile_start[blockIdx.x];
    for (int k = 0; k < ASSIGN_TILE / DXO_BLOCK; ++k) {
        const int64_t e = base + k * DXO_BLOCK + threadIdx.x;
        const bool win = assign_wins(dofs, owner, e, n, coeff_size);
        const unsigned long long m = __ballot(win);
        if (lane == 0) part[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < DXO_BLOCK / DXO_WAVE; ++w) {
            if (w < wave) before += part[w];
            all += part[w];
        }
        if (win) {
            const uint32_t pos = at + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            src_s[pos] = (I)assign_src(a, e);
            dst_s[pos] = dofs[e];
        }
        at += all;
        __syncthreads();
    }
}

template <typename I, typename T>
__global__ __launch_bounds__(DXO_BLOCK) void assign_apply_pairs(const I* __restrict__ src_s, const int32_t* __restrict__ dst_s,
                                                                const T* __restrict__ values, T* __restrict__ coeff, int64_t n_pairs) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += stride) coeff[dst_s[i]] = values[src_s[i]];
}

}  // namespace

// element width of a descriptor: 0 (a caller of ABI version 1, where the field was padding it zeroed) means 8
static int assign_elem_bytes(const dxo_assign_desc* d) { return d->elem_bytes == 0 ? 8 : d->elem_bytes; }
static bool assign_elem_ok(int eb) { return eb == 4 || eb == 8 || eb == 16; }

// Scratch of the owner pass: [error word (8 bytes)][owner table: coeff_size words of 4 or 8 bytes], zeroed on the stream.
struct OwnerTable {
    unsigned long long* bad = nullptr;
    void* words = nullptr;
    bool narrow = true;
};
static bool assign_owner_table(dxo_ctx* ctx, hipStream_t s, int64_t n_entries, int64_t coeff_size, OwnerTable* t) {
    t->narrow = n_entries < 0xffffffffLL && ctx->assign_owner_bits != 64;
    const size_t need = 8 + (size_t)coeff_size * (t->narrow ? 4 : 8);
    char* base = static_cast<char*>(dxo_scratch(ctx, s, need));
    if (!base) return false;
    t->bad = reinterpret_cast<unsigned long long*>(base);
    t->words = base + 8;
    return hipMemsetAsync(base, 0, need, s) == hipSuccess;
}
static void assign_owner_launch(const OwnerTable& t, const AssignDev& a, const int32_t* flat_dofs, int64_t coeff_size, int blocks, hipStream_t s) {
    if (t.narrow) hipLaunchKernelGGL(assign_owner<uint32_t>, dim3(blocks), dim3(DXO_BLOCK), 0, s, a, flat_dofs, (uint32_t*)t.words, t.bad, coeff_size);
    else          hipLaunchKernelGGL(assign_owner<unsigned long long>, dim3(blocks), dim3(DXO_BLOCK), 0, s, a, flat_dofs, (unsigned long long*)t.words, t.bad, coeff_size);
}

struct dxo_assign_plan {
    int64_t coeff_size = 0, n_values = 0;
    int elem_bytes = 8;
    bool wide = false;          // int64 source positions (values array beyond 2^31 entries)
    void* src = nullptr;        // device: int32 or int64 per coefficient entry
    // the same plan in source order: n_pairs winners, (position in values, coefficient entry) each; absent for small plans and when its block
    // could not be had
    int64_t n_pairs = 0;
    void* src_s = nullptr;
    int32_t* dst_s = nullptr;
    // which form dxo_assign_apply launches: 0 = not decided yet (the first call times both on the caller's arrays), 1 = dof order, 2 = source order
    mutable int form = 1;
    mutable float form_ms[2] = {0.0f, 0.0f};
};
constexpr int64_t ASSIGN_PAIRS_MIN = 1 << 20;      // coefficient entries below which a plan keeps the dof-order form only (the launch bounds those)

extern "C" int dxo_assign_plan_create(dxo_ctx* ctx, const dxo_assign_desc* d, const int32_t* flat_dofs, int64_t coeff_size,
                                      dxo_assign_plan** out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!d || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign_plan_create: NULL argument");
    *out = nullptr;
    if (d->n_cells < 0 || d->n_pts < 1 || d->val_size < 1 || d->offset < 0 || d->comp_size < d->val_size ||
        d->n_points_total < d->offset + d->n_pts || coeff_size < 0)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_assign_plan_create: inconsistent sizes");
    if (!assign_elem_ok(assign_elem_bytes(d))) return dxo_fail(ctx, DXO_E_DIM, "dxo_assign_plan_create: elem_bytes must be 4, 8 or 16 (0 = 8)");
    const int64_t n = d->n_cells * d->n_pts * d->val_size;
    if (n > 0 && !flat_dofs) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign_plan_create: flat_dofs is NULL");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    dxo_assign_plan* pl = new dxo_assign_plan;
    pl->coeff_size = coeff_size;
    pl->elem_bytes = assign_elem_bytes(d);
    pl->n_values = d->n_cells * (int64_t)d->n_points_total * d->comp_size;
    pl->wide = pl->n_values > 0x7fffffffLL;
    const size_t sb = (size_t)(coeff_size ? coeff_size : 1) * (pl->wide ? 8 : 4);
    if (hipMalloc(&pl->src, sb) != hipSuccess) {
        (void)hipGetLastError();
        delete pl;
        return dxo_hip_fail(ctx, hipErrorOutOfMemory, "dxo_assign_plan_create: plan allocation");
    }
    OwnerTable t;
    auto bail = [&](int rc) { (void)hipFree(pl->src); delete pl; return rc; };
    if (!assign_owner_table(ctx, s, n, coeff_size, &t)) return bail(dxo_hip_fail(ctx, hipErrorOutOfMemory, "dxo_assign_plan_create: owner table"));
    AssignDev a{d->n_cells, d->n_pts, d->val_size, d->offset, d->n_points_total, d->comp_size};
    const int64_t cap = (int64_t)ctx->compute_units * 16;
    if (n > 0) {
        int64_t blocks = (n + DXO_BLOCK - 1) / DXO_BLOCK;
        if (blocks > cap) blocks = cap;
        assign_owner_launch(t, a, flat_dofs, coeff_size, (int)blocks, s);
    }
    if (coeff_size > 0) {
        int64_t blocks = (coeff_size + DXO_BLOCK - 1) / DXO_BLOCK;
        if (blocks > cap) blocks = cap;
#define DXO_FINISH(W, I) hipLaunchKernelGGL((assign_plan_finish<W, I>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, a, (const W*)t.words, (I*)pl->src, coeff_size)
        if (t.narrow) { if (pl->wide) DXO_FINISH(uint32_t, int64_t); else DXO_FINISH(uint32_t, int32_t); }
        else          { if (pl->wide) DXO_FINISH(unsigned long long, int64_t); else DXO_FINISH(unsigned long long, int32_t); }
#undef DXO_FINISH
    }
    // the source-order form (option "assign_plan_form": 0 = both, the first apply chooses; 1 = dof order only; 2 = source order)
    uint32_t* tiles = nullptr;
    unsigned long long* d_total = nullptr;
    const int64_t n_tiles = (n + ASSIGN_TILE - 1) / ASSIGN_TILE;
    const bool want_pairs = ctx->assign_plan_form != 1 && n > 0 && n_tiles < 0x7fffffffLL && coeff_size < 0xffffffffLL &&
                            (coeff_size >= ASSIGN_PAIRS_MIN || ctx->assign_plan_form == 2);
    if (want_pairs) {
        // tile counts + the total live in a block of their own (dxo_scratch holds the owner table)
        if (hipMalloc(&tiles, ((size_t)n_tiles + 2) * 4 + 8) != hipSuccess) { (void)hipGetLastError(); tiles = nullptr; }
        if (tiles) {
            d_total = reinterpret_cast<unsigned long long*>(tiles + ((n_tiles + 1) & ~1LL));
            if (t.narrow) hipLaunchKernelGGL(assign_tile_count<uint32_t>, dim3((int)n_tiles), dim3(DXO_BLOCK), 0, s, n, flat_dofs, (const uint32_t*)t.words, coeff_size, tiles);
            else          hipLaunchKernelGGL(assign_tile_count<unsigned long long>, dim3((int)n_tiles), dim3(DXO_BLOCK), 0, s, n, flat_dofs, (const unsigned long long*)t.words, coeff_size, tiles);
            hipLaunchKernelGGL(assign_tile_scan, dim3(1), dim3(1024), 0, s, tiles, n_tiles, d_total);
        }
    }
    unsigned long long bad = 0, total = 0;
    bool sync_ok = hipMemcpyAsync(&bad, t.bad, sizeof bad, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (sync_ok && tiles) sync_ok = hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, s) == hipSuccess;
    if (!sync_ok || hipStreamSynchronize(s) != hipSuccess) {
        if (tiles) (void)hipFree(tiles);
        return bail(dxo_hip_fail(ctx, hipGetLastError(), "dxo_assign_plan_create: synchronisation"));
    }
    if (tiles && !bad && total > 0) {
        const size_t ib = pl->wide ? 8 : 4;
        if (hipMalloc(&pl->src_s, (size_t)total * ib) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&pl->dst_s), (size_t)total * 4) != hipSuccess) {
            (void)hipGetLastError();      // no room for the second form: the plan keeps the first
            if (pl->src_s) (void)hipFree(pl->src_s);
            pl->src_s = nullptr;
            pl->dst_s = nullptr;
        } else {
            pl->n_pairs = (int64_t)total;
#define DXO_PAIRS(W, I) hipLaunchKernelGGL((assign_tile_pairs<W, I>), dim3((int)n_tiles), dim3(DXO_BLOCK), 0, s, a, n, flat_dofs, (const W*)t.words, coeff_size, tiles, (I*)pl->src_s, pl->dst_s)
            if (t.narrow) { if (pl->wide) DXO_PAIRS(uint32_t, int64_t); else DXO_PAIRS(uint32_t, int32_t); }
            else          { if (pl->wide) DXO_PAIRS(unsigned long long, int64_t); else DXO_PAIRS(unsigned long long, int32_t); }
#undef DXO_PAIRS
            if (hipStreamSynchronize(s) != hipSuccess) {      // flat_dofs is not kept: the pairs are complete when this call returns
                (void)hipFree(tiles);
                (void)hipFree(pl->src_s);
                (void)hipFree(pl->dst_s);
                return bail(dxo_hip_fail(ctx, hipGetLastError(), "dxo_assign_plan_create: source-order form"));
            }
            pl->form = ctx->assign_plan_form == 2 ? 2 : 0;
        }
    }
    if (tiles) (void)hipFree(tiles);
    if (bad) {   // the NumPy assigner raises IndexError (external_operator.py:287); no plan is made
        char msg[160];
        std::snprintf(msg, sizeof msg, "dxo_assign_plan_create: %llu flat_dofs entries outside [0, coeff_size = %lld)", bad, (long long)coeff_size);
        return bail(dxo_fail(ctx, DXO_E_SIZE, msg));
    }
    *out = pl;
    return DXO_OK;
}

extern "C" void dxo_assign_plan_destroy(dxo_ctx* ctx, dxo_assign_plan* plan) {
    if (!plan) return;
    DXO_LOCK(ctx);
    if (plan->src) (void)hipFree(plan->src);
    if (plan->src_s) (void)hipFree(plan->src_s);
    if (plan->dst_s) (void)hipFree(plan->dst_s);
    delete plan;
}

extern "C" int dxo_assign_apply(dxo_ctx* ctx, const dxo_assign_plan* plan, const void* values, void* coeff) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!plan) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign_apply: plan is NULL");
    if (plan->coeff_size == 0) return DXO_OK;
    if (!values || !coeff) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign_apply: NULL array");
    if (((uintptr_t)values | (uintptr_t)coeff) & (uintptr_t)(plan->elem_bytes - 1))
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_assign_apply: values and coeff must be aligned to the element width");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    // about one element per thread: both forms are a dependent index -> value -> store chain per element, and a grid-stride loop of ten turns on
    // 16 workgroups per CU hides less of it (108^3 Q2 hexahedra: 0.108 -> 0.086 / 0.090 -> 0.081 ms, scripts/exp/assign_sorted_probe.hip)
    const int64_t cap = (int64_t)ctx->compute_units * 128;
    auto launch = [&](int form) {
        const int64_t work = form == 2 ? plan->n_pairs : plan->coeff_size;
        int64_t blocks = (work + DXO_BLOCK - 1) / DXO_BLOCK;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) return;
#define DXO_APPLY(I, T) do { if (form == 2) hipLaunchKernelGGL((assign_apply_pairs<I, T>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, (const I*)plan->src_s, plan->dst_s, (const T*)values, (T*)coeff, plan->n_pairs); \
                             else hipLaunchKernelGGL((assign_apply<I, T>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, (const I*)plan->src, (const T*)values, (T*)coeff, plan->coeff_size); } while (0)
        if (plan->wide) {
            if (plan->elem_bytes == 4) DXO_APPLY(int64_t, uint32_t); else if (plan->elem_bytes == 8) DXO_APPLY(int64_t, unsigned long long); else DXO_APPLY(int64_t, assign_u128);
        } else {
            if (plan->elem_bytes == 4) DXO_APPLY(int32_t, uint32_t); else if (plan->elem_bytes == 8) DXO_APPLY(int32_t, unsigned long long); else DXO_APPLY(int32_t, assign_u128);
        }
#undef DXO_APPLY
    };
    if (plan->form == 0) {
        // first use of a plan that carries both forms: each is launched twice on the caller's own arrays (the assignment is idempotent: every launch
        // leaves the same coefficient) and the second launch of each is timed. A stream that is being captured into a graph cannot be waited on:
        // such a call takes the dof-order form and leaves the choice to a later one.
        hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap_st) != hipSuccess) { (void)hipGetLastError(); cap_st = hipStreamCaptureStatusNone; }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (cap_st == hipStreamCaptureStatusNone && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
            bool timed = true;
            for (int f = 1; f <= 2 && timed; ++f) {
                launch(f);
                timed = hipEventRecord(e0, s) == hipSuccess;
                launch(f);
                timed = timed && hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
                        hipEventElapsedTime(&plan->form_ms[f - 1], e0, e1) == hipSuccess;
            }
            if (timed) plan->form = plan->form_ms[1] < plan->form_ms[0] ? 2 : 1;
            else (void)hipGetLastError();
        } else {
            launch(1);
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return dxo_device_end(ctx, s);
    }
    launch(plan->form);
    return dxo_device_end(ctx, s);
}

extern "C" int dxo_assign_plan_form(const dxo_assign_plan* plan, double* ms_dof_order, double* ms_source_order) {
    if (!plan) return DXO_E_NULL;
    if (ms_dof_order) *ms_dof_order = plan->form_ms[0];
    if (ms_source_order) *ms_source_order = plan->form_ms[1];
    return plan->form;
}

extern "C" int dxo_assign(dxo_ctx* ctx, const dxo_assign_desc* d, const int32_t* flat_dofs, const void* values,
                          void* coeff, int64_t coeff_size) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!d) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign: descriptor is NULL");
    if (d->n_cells < 0 || d->n_pts < 1 || d->val_size < 1 || d->offset < 0 || d->comp_size < d->val_size ||
        d->n_points_total < d->offset + d->n_pts || coeff_size < 0)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_assign: inconsistent sizes");
    const int eb = assign_elem_bytes(d);
    if (!assign_elem_ok(eb)) return dxo_fail(ctx, DXO_E_DIM, "dxo_assign: elem_bytes must be 4, 8 or 16 (0 = 8)");
    const int64_t n = d->n_cells * d->n_pts * d->val_size;
    if (n == 0) return DXO_OK;
    if (!flat_dofs || !values || !coeff) return dxo_fail(ctx, DXO_E_NULL, "dxo_assign: NULL array");
    if (((uintptr_t)values | (uintptr_t)coeff) & (uintptr_t)(eb - 1))
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_assign: values and coeff must be aligned to the element width");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    // owner table: one word per coefficient entry behind the error word, in the context's device-path scratch
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    OwnerTable t;
    if (!assign_owner_table(ctx, s, n, coeff_size, &t)) return dxo_hip_fail(ctx, hipErrorOutOfMemory, "dxo_assign: owner table");
    AssignDev a{d->n_cells, d->n_pts, d->val_size, d->offset, d->n_points_total, d->comp_size};
    int64_t blocks = (n + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * (DXO_ASSIGN_CAP);
    if (blocks > cap) blocks = cap;
    assign_owner_launch(t, a, flat_dofs, coeff_size, (int)blocks, s);
#define DXO_STORE(W, T) hipLaunchKernelGGL((assign_store<W, T>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, a, flat_dofs, (const W*)t.words, (const T*)values, (T*)coeff, coeff_size)
    if (t.narrow) { if (eb == 4) DXO_STORE(uint32_t, uint32_t); else if (eb == 8) DXO_STORE(uint32_t, unsigned long long); else DXO_STORE(uint32_t, assign_u128); }
    else          { if (eb == 4) DXO_STORE(unsigned long long, uint32_t); else if (eb == 8) DXO_STORE(unsigned long long, unsigned long long); else DXO_STORE(unsigned long long, assign_u128); }
#undef DXO_STORE
    rc = dxo_device_end(ctx, s);
    if (rc != DXO_OK || !ctx->assign_validate) return rc;
    unsigned long long bad = 0;
    DXO_HIP(ctx, hipMemcpyAsync(&bad, t.bad, sizeof bad, hipMemcpyDeviceToHost, s));
    DXO_HIP(ctx, hipStreamSynchronize(s));
    if (bad) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "dxo_assign: %llu flat_dofs entries outside [0, coeff_size = %lld)", bad, (long long)coeff_size);
        return dxo_fail(ctx, DXO_E_SIZE, msg);
    }
    return DXO_OK;
}
