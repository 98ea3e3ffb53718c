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
};

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
    unsigned long long bad = 0;
    if (hipMemcpyAsync(&bad, t.bad, sizeof bad, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return bail(dxo_hip_fail(ctx, hipGetLastError(), "dxo_assign_plan_create: synchronisation"));
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
    int64_t blocks = (plan->coeff_size + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * 16;
    if (blocks > cap) blocks = cap;
#define DXO_APPLY(I, T) hipLaunchKernelGGL((assign_apply<I, T>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, (const I*)plan->src, (const T*)values, (T*)coeff, plan->coeff_size)
    if (plan->wide) {
        if (plan->elem_bytes == 4) DXO_APPLY(int64_t, uint32_t); else if (plan->elem_bytes == 8) DXO_APPLY(int64_t, unsigned long long); else DXO_APPLY(int64_t, assign_u128);
    } else {
        if (plan->elem_bytes == 4) DXO_APPLY(int32_t, uint32_t); else if (plan->elem_bytes == 8) DXO_APPLY(int32_t, unsigned long long); else DXO_APPLY(int32_t, assign_u128);
    }
#undef DXO_APPLY
    return dxo_device_end(ctx, s);
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
    const int64_t cap = (int64_t)ctx->compute_units * 16;
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
