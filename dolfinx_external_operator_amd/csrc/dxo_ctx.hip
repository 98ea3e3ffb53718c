// dxo_ctx.hip — context, options, pinned host memory, the chunked host pipeline.
// Host-side runtime of libdxo_hip.so; the kernels live in von_mises.hip / heat.hip / ...
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

#include "dxo_common.h"
#include "host_pool.h"

// ---------------------------------------------------------------------------------------------- host worker threads
void dxo_host_parallel_for(dxo_ctx* c, int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& fn) {
    dxo_pool_parallel_for(c->pool, (int)c->host_threads, n, grain, fn);
}

namespace {

struct SlotEvents {
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};  // h2d start, kernel start, kernel end, d2h end
    bool used = false;
};

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Page-locked host ranges this library has handed out or registered (dxo_host_alloc, dxo_host_register), process-wide: the
// small-batch path of the host pipeline lets the kernel read / write such arrays IN PLACE (they are device-mapped) instead of
// copying them through its staging block. Keyed by base address.
// (heap objects that are never destroyed: a page-locked block may be released by a finaliser that runs while the process is exiting,
// after the destructors of this library's statics)
std::mutex& g_pinned_mu = *new std::mutex;
std::map<uintptr_t, size_t>& g_pinned = *new std::map<uintptr_t, size_t>;

}  // namespace

void dxo_pinned_note(void* base, size_t bytes, bool add) {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    if (add) g_pinned[(uintptr_t)base] = bytes;
    else g_pinned.erase((uintptr_t)base);
}

// device-side address of [p, p + bytes) when that range lies inside a page-locked block of the registry, else nullptr
void* dxo_pinned_mapped(const void* p, size_t bytes) {
    if (!p) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.upper_bound((uintptr_t)p);
        if (it == g_pinned.begin()) return nullptr;
        --it;
        if ((uintptr_t)p + bytes > it->first + it->second) return nullptr;
    }
    void* m = nullptr;
    if (hipHostGetDevicePointer(&m, const_cast<void*>(p), 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return m;
}

extern "C" {

int dxo_abi_version(void) { return DXO_ABI_VERSION; }

int dxo_device_count(int* count) {
    if (!count) return DXO_E_NULL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        (void)hipGetLastError();
        return DXO_E_NODEVICE;
    }
    *count = n;
    return DXO_OK;
}

int dxo_ctx_create(int device, dxo_ctx** out) {
    if (!out) return DXO_E_NULL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        (void)hipGetLastError();
        return DXO_E_NODEVICE;
    }
    dxo_ctx* c = new dxo_ctx();
    c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < DXO_HOST_SLOTS && e == hipSuccess; ++i)
        e = hipStreamCreateWithFlags(&c->slot_stream[i], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_stop);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        int code = dxo_hip_fail(nullptr, e, "dxo_ctx_create");
        dxo_ctx_destroy(c);
        return code;
    }
    c->compute_units = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = c;
    return DXO_OK;
}

int dxo_ctx_destroy(dxo_ctx* c) {
    if (!c) return DXO_E_NULL;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < DXO_HOST_SLOTS; ++i) {
        if (c->slot_buf[i]) (void)hipFree(c->slot_buf[i]);
        if (c->slot_stream[i]) (void)hipStreamDestroy(c->slot_stream[i]);
        if (i == 0) {
            if (c->small_pinned) (void)hipHostFree(c->small_pinned);
            for (auto& e : c->small_ev)
                if (e) (void)hipEventDestroy(e);
        }
    }
    for (int i = 0; i <= DXO_HOST_SLOTS; ++i)
    {
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
        if (c->stage[i]) (void)hipFree(c->stage[i]);
    }
    dxo_arena_release_all(c);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (auto& e : c->cal_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    dxo_host_pool_destroy(c->pool);
    delete c;
    return DXO_OK;
}

const char* dxo_last_error(const dxo_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int dxo_ctx_device_info(dxo_ctx* c, dxo_device_info* info) {
    if (!c || !info) return DXO_E_NULL;
    DXO_LOCK(c);
    hipDeviceProp_t prop;
    DXO_HIP(c, hipGetDeviceProperties(&prop, c->device));
    std::memset(info, 0, sizeof *info);
    std::strncpy(info->name, prop.name, sizeof info->name - 1);
    std::strncpy(info->arch, prop.gcnArchName, sizeof info->arch - 1);
    info->compute_units = prop.multiProcessorCount;
    info->wavefront_size = prop.warpSize;
    info->total_mem_bytes = (int64_t)prop.totalGlobalMem;
    return DXO_OK;
}

int dxo_ctx_set_stream(dxo_ctx* c, void* hip_stream) {
    if (!c) return DXO_E_NULL;
    DXO_LOCK(c);
    c->user_stream = (hipStream_t)hip_stream;
    c->use_user_stream = true;  // NULL selects the legacy default stream explicitly
    return DXO_OK;
}

int dxo_ctx_synchronize(dxo_ctx* c) {
    if (!c) return DXO_E_NULL;
    DXO_LOCK(c);
    DXO_HIP(c, hipSetDevice(c->device));
    DXO_HIP(c, hipStreamSynchronize(dxo_launch_stream(c)));
    for (int i = 0; i < DXO_HOST_SLOTS; ++i) DXO_HIP(c, hipStreamSynchronize(c->slot_stream[i]));
    return DXO_OK;
}

static int64_t* option_slot(dxo_ctx* c, const char* key) {
    if (!std::strcmp(key, "vm_variant")) return &c->vm_variant;
    if (!std::strcmp(key, "host_chunk_points")) return &c->host_chunk_points;
    if (!std::strcmp(key, "nontemporal")) return &c->nontemporal;
    if (!std::strcmp(key, "timing")) return &c->timing;
    if (!std::strcmp(key, "blocks_per_cu")) return &c->blocks_per_cu;
    if (!std::strcmp(key, "mc_variant")) return &c->mc_variant;
    if (!std::strcmp(key, "mc_blocks_per_cu")) return &c->mc_blocks_per_cu;
    if (!std::strcmp(key, "mc_waves_per_simd")) return &c->mc_waves_per_simd;
    if (!std::strcmp(key, "icnn_variant")) return &c->icnn_variant;
    if (!std::strcmp(key, "adjoint_atomics")) return &c->adjoint_atomics;
    if (!std::strcmp(key, "adjoint_patch")) return &c->adjoint_patch;
    if (!std::strcmp(key, "adjoint_mfma")) return &c->adjoint_mfma;
    if (!std::strcmp(key, "mgpu_chunks")) return &c->mgpu_chunks;
    if (!std::strcmp(key, "adjoint_cell")) return &c->adjoint_cell;
    if (!std::strcmp(key, "vm_residual_fused")) return &c->vm_residual_fused;
    if (!std::strcmp(key, "consumer_overwrite")) return &c->consumer_overwrite;
    if (!std::strcmp(key, "operand_cell")) return &c->operand_cell;
    if (!std::strcmp(key, "mc_part_points")) return &c->mc_part_points;
    if (!std::strcmp(key, "host_small_bytes")) return &c->host_small_bytes;
    if (!std::strcmp(key, "host_zero_copy_bytes")) return &c->host_zero_copy_bytes;
    if (!std::strcmp(key, "host_zero_copy_piece_bytes")) return &c->host_zero_copy_piece_bytes;
    if (!std::strcmp(key, "vm_host_tangent")) return &c->vm_host_tangent;
    if (!std::strcmp(key, "vm_mark_indeterminate")) return &c->vm_mark_indeterminate;
    if (!std::strcmp(key, "assign_validate")) return &c->assign_validate;
    if (!std::strcmp(key, "assign_owner_bits")) return &c->assign_owner_bits;
    if (!std::strcmp(key, "assign_plan_form")) return &c->assign_plan_form;
    if (!std::strcmp(key, "placement_mode")) return &c->placement_mode;
    if (!std::strcmp(key, "placement_candidates")) return &c->placement_candidates;
    if (!std::strcmp(key, "placement_min_bytes")) return &c->placement_min_bytes;
    if (!std::strcmp(key, "placement_good_GBps")) return &c->placement_good_GBps;
    if (!std::strcmp(key, "placement_vmm")) return &c->placement_vmm;
    if (!std::strcmp(key, "placement_rounds")) return &c->placement_rounds;
    if (!std::strcmp(key, "placement_cache")) return &c->placement_cache;
    if (!std::strcmp(key, "placement_accept_pct")) return &c->placement_accept_pct;
    if (!std::strcmp(key, "placement_standout_pct")) return &c->placement_standout_pct;
    if (!std::strcmp(key, "placement_probe")) return &c->placement_probe;
    if (!std::strcmp(key, "placement_good_mix_GBps")) return &c->placement_good_mix_GBps;
    if (!std::strcmp(key, "host_threads")) return &c->host_threads;
    if (!std::strcmp(key, "vm_rebuild_chunk_points")) return &c->vm_rebuild_chunk_points;
    if (!std::strcmp(key, "vm_rebuild_min_points")) return &c->vm_rebuild_min_points;
    return nullptr;
}

int dxo_ctx_set_option(dxo_ctx* c, const char* key, int64_t value) {
    if (!c || !key) return DXO_E_NULL;
    DXO_LOCK(c);
    int64_t* slot = option_slot(c, key);
    if (!slot) return dxo_fail(c, DXO_E_OPTION, "unknown option");
    if (value < 0) return dxo_fail(c, DXO_E_OPTION, "option value must be >= 0");
    if (slot == &c->icnn_variant && value > dxo_icnn_variant_max())
        return dxo_fail(c, DXO_E_OPTION, "icnn_variant: 0 (VALU), 1 (fp32 MFMA), 2 (split-bf16 MFMA); 3 and 4 exist only in a -DDXO_EXPERIMENTS build (scripts/exp/icnn_variants.h)");
    if (slot == &c->adjoint_patch && value != 0 && !dxo_adjoint_patch_available())
        return dxo_fail(c, DXO_E_OPTION, "adjoint_patch: the patch form exists only in a -DDXO_EXPERIMENTS build (scripts/exp/adjoint_patch.h)");
    if (slot == &c->assign_plan_form && value > 2) return dxo_fail(c, DXO_E_OPTION, "assign_plan_form: 0 (both, the first apply chooses), 1 (dof order), 2 (source order)");
    if (slot == &c->assign_owner_bits && value != 0 && value != 64) return dxo_fail(c, DXO_E_OPTION, "assign_owner_bits: 0 (32-bit while the entry count fits) or 64");
    if (slot == &c->mc_blocks_per_cu && value < 1) return dxo_fail(c, DXO_E_OPTION, "mc_blocks_per_cu < 1");
    if (slot == &c->vm_rebuild_chunk_points && value < DXO_WAVE) return dxo_fail(c, DXO_E_OPTION, "vm_rebuild_chunk_points < 64");
    if (slot == &c->host_chunk_points && value < DXO_WAVE) return dxo_fail(c, DXO_E_OPTION, "host_chunk_points < 64");
    *slot = value;
    return DXO_OK;
}

int dxo_ctx_get_option(dxo_ctx* c, const char* key, int64_t* value) {
    if (!c || !key || !value) return DXO_E_NULL;
    DXO_LOCK(c);
    if (!std::strcmp(key, "device")) {     // read-only: the HIP device ordinal this context was created on
        *value = c->device;
        return DXO_OK;
    }
    int64_t* slot = option_slot(c, key);
    if (!slot) return dxo_fail(c, DXO_E_OPTION, "unknown option");
    *value = *slot;
    return DXO_OK;
}

int dxo_last_timing(dxo_ctx* c, dxo_timing* t) {
    if (!c || !t) return DXO_E_NULL;
    DXO_LOCK(c);
    if (c->ev_pending) {
        DXO_HIP(c, hipEventSynchronize(c->ev_stop));
        float ms = 0.f;
        DXO_HIP(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
        c->last = {0.0, (double)ms, 0.0, (double)ms};
        c->ev_pending = false;
    }
    *t = c->last;
    return DXO_OK;
}

// ctx may be NULL for both (page-locked host memory belongs to the process, not to a device context): a buffer can
// then outlive the context it was first used with, which is what the Python binding's recycling pool needs.
int dxo_host_alloc(dxo_ctx* c, int64_t bytes, void** ptr) {
    DXO_LOCK(c);
    if (!ptr) return DXO_E_NULL;
    if (bytes < 0) return dxo_fail(c, DXO_E_SIZE, "negative size");
    *ptr = nullptr;
    if (c) DXO_HIP(c, hipSetDevice(c->device));
    DXO_HIP(c, hipHostMalloc(ptr, bytes > 0 ? (size_t)bytes : 1, hipHostMallocDefault));
    dxo_pinned_note(*ptr, bytes > 0 ? (size_t)bytes : 1, true);
    return DXO_OK;
}

// Page-lock memory the CALLER owns (e.g. the storage of a DOLFINx coefficient the results are written into at every call):
// DMA then goes straight to it instead of through the driver's staging buffers.
int dxo_host_register(dxo_ctx* c, void* ptr, int64_t bytes) {
    DXO_LOCK(c);
    if (!ptr) return DXO_E_NULL;
    if (bytes <= 0) return dxo_fail(c, DXO_E_SIZE, "dxo_host_register: size must be positive");
    if (c) DXO_HIP(c, hipSetDevice(c->device));
    DXO_HIP(c, hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    dxo_pinned_note(ptr, (size_t)bytes, true);
    return DXO_OK;
}

int dxo_host_unregister(dxo_ctx* c, void* ptr) {
    DXO_LOCK(c);
    if (!ptr) return DXO_OK;
    dxo_pinned_note(ptr, 0, false);
    DXO_HIP(c, hipHostUnregister(ptr));
    return DXO_OK;
}

int dxo_host_free(dxo_ctx* c, void* ptr) {
    DXO_LOCK(c);
    if (!ptr) return DXO_OK;
    dxo_pinned_note(ptr, 0, false);
    DXO_HIP(c, hipHostFree(ptr));
    return DXO_OK;
}

int dxo_device_alloc(dxo_ctx* c, int64_t bytes, void** ptr) {
    if (!c || !ptr) return DXO_E_NULL;
    DXO_LOCK(c);
    if (bytes < 0) return dxo_fail(c, DXO_E_SIZE, "negative size");
    *ptr = nullptr;
    DXO_HIP(c, hipSetDevice(c->device));
    if (hipMalloc(ptr, bytes > 0 ? (size_t)bytes : 1) != hipSuccess && !c->arena_cache.empty()) {
        (void)hipGetLastError();
        dxo_arena_cache_drop(c);      // the block dxo_output_free retained gives way to a request that would otherwise fail
        *ptr = nullptr;
    }
    if (!*ptr) DXO_HIP(c, hipMalloc(ptr, bytes > 0 ? (size_t)bytes : 1));
    return DXO_OK;
}

int dxo_device_free(dxo_ctx* c, void* ptr) {
    if (!c) return DXO_E_NULL;
    DXO_LOCK(c);
    if (!ptr) return DXO_OK;
    DXO_HIP(c, hipSetDevice(c->device));
    DXO_HIP(c, hipFree(ptr));
    return DXO_OK;
}

int dxo_copy(dxo_ctx* c, void* dst, const void* src, int64_t bytes, int kind) {
    if (!c) return DXO_E_NULL;
    DXO_LOCK(c);
    if (bytes < 0) return dxo_fail(c, DXO_E_SIZE, "negative size");
    if (kind < 0 || kind > 2) return dxo_fail(c, DXO_E_MEM, "dxo_copy: kind must be 0 (H2D), 1 (D2H) or 2 (D2D)");
    if (bytes == 0) return DXO_OK;
    if (!dst || !src) return dxo_fail(c, DXO_E_NULL, "dxo_copy: NULL pointer");
    DXO_HIP(c, hipSetDevice(c->device));
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    hipStream_t s = dxo_launch_stream(c);
    DXO_HIP(c, hipMemcpyAsync(dst, src, (size_t)bytes, k, s));
    DXO_HIP(c, hipStreamSynchronize(s));
    return DXO_OK;
}

}  // extern "C"

int dxo_device_begin(dxo_ctx* c, hipStream_t s) {
    DXO_HIP(c, hipSetDevice(c->device));
    if (c->timing) DXO_HIP(c, hipEventRecord(c->ev_start, s));
    return DXO_OK;
}

int dxo_device_end(dxo_ctx* c, hipStream_t s) {
    DXO_HIP(c, hipGetLastError());
    if (c->timing) {
        DXO_HIP(c, hipEventRecord(c->ev_stop, s));
        c->ev_pending = true;
    }
    return DXO_OK;
}

void* dxo_scratch(dxo_ctx* ctx, hipStream_t s, size_t bytes) {
    int slot = DXO_HOST_SLOTS;
    for (int i = 0; i < DXO_HOST_SLOTS; ++i)
        if (ctx->slot_stream[i] == s) slot = i;
    if (slot == DXO_HOST_SLOTS) {
        if (ctx->scratch_stream_set && ctx->scratch_stream != s && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) {
            (void)hipGetLastError();   // the earlier stream may have been destroyed by its owner: nothing left to wait for
        }
        ctx->scratch_stream = s;
        ctx->scratch_stream_set = true;
    }
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
            (void)hipFree(ctx->scratch[slot]);
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&ctx->scratch[slot], want) != hipSuccess) return nullptr;
        ctx->scratch_bytes[slot] = want;
    }
    return ctx->scratch[slot];
}

void* dxo_stage(dxo_ctx* ctx, hipStream_t s, size_t bytes) {
    int slot = DXO_HOST_SLOTS;
    for (int i = 0; i < DXO_HOST_SLOTS; ++i)
        if (ctx->slot_stream[i] == s) slot = i;
    if (slot == DXO_HOST_SLOTS) {   // device-path buffer: same hand-over between caller streams as dxo_scratch
        if (ctx->scratch_stream_set && ctx->scratch_stream != s && hipStreamSynchronize(ctx->scratch_stream) != hipSuccess) (void)hipGetLastError();
        ctx->scratch_stream = s;
        ctx->scratch_stream_set = true;
    }
    if (ctx->stage_bytes[slot] < bytes) {
        if (ctx->stage[slot]) {
            if (hipStreamSynchronize(s) != hipSuccess) return nullptr;
            (void)hipFree(ctx->stage[slot]);
            ctx->stage[slot] = nullptr;
            ctx->stage_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipMalloc(&ctx->stage[slot], want) != hipSuccess) return nullptr;
        ctx->stage_bytes[slot] = want;
    }
    return ctx->stage[slot];
}

int dxo_grid_for_tiles(const dxo_ctx* c, int64_t n_tiles, int tiles_per_block) {
    int64_t blocks = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    if (blocks < 1) blocks = 1;
    if (c->blocks_per_cu > 0) {
        int64_t cap = (int64_t)c->compute_units * c->blocks_per_cu;
        if (blocks > cap) blocks = cap;
    }
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
    return (int)blocks;
}

int dxo_run_host_pipeline(dxo_ctx* c, int64_t n, const std::vector<dxo_span>& inputs,
                          const std::vector<dxo_span>& outputs, dxo_chunk_launch launch, void* user,
                          int64_t points_per_unit, dxo_chunk_post post, bool stream_once) {
    DXO_HIP(c, hipSetDevice(c->device));
    c->last = {0, 0, 0, 0};
    c->ev_pending = false;
    if (n == 0) return DXO_OK;
    {
        // ---- small batches (the reference's demo meshes: hundreds to thousands of points): the fixed cost of a call is
        // the number of driver calls, not bytes. Inputs are packed into ONE pinned staging buffer and go up in one copy,
        // the outputs come back in one copy; events are created once per context.
        size_t in_bytes = 0, out_bytes = 0;
        for (const auto& s : inputs)
            if (!s.dev) in_bytes += round_up(s.bytes_pp * (size_t)n, 256);
        for (const auto& s : outputs) out_bytes += round_up(s.bytes_pp * (size_t)n, 256);
        // (a batch the caller's chunk size would split — option host_chunk_points below the batch — takes the chunked ring it asks for)
        if ((int64_t)(in_bytes + out_bytes) <= c->host_small_bytes && n * (points_per_unit > 0 ? points_per_unit : 1) <= c->host_chunk_points) {
            const size_t need = in_bytes + out_bytes;
            if (need > c->small_pinned_bytes) {
                if (c->small_pinned) DXO_HIP(c, hipHostFree(c->small_pinned));
                c->small_pinned = nullptr;
                c->small_pinned_bytes = 0;
                const size_t want = need > (size_t)c->host_small_bytes ? need : (size_t)c->host_small_bytes;
                DXO_HIP(c, hipHostMalloc(&c->small_pinned, want, hipHostMallocDefault));
                c->small_pinned_bytes = want;
            }
            // device side: slot 0's buffer (grown like the chunked path grows it)
            if (need > c->slot_bytes) {
                for (int i = 0; i < DXO_HOST_SLOTS; ++i) {
                    DXO_HIP(c, hipStreamSynchronize(c->slot_stream[i]));
                    if (c->slot_buf[i]) DXO_HIP(c, hipFree(c->slot_buf[i]));
                    c->slot_buf[i] = nullptr;
                }
                c->slot_bytes = 0;
                const size_t want = need > (size_t)c->host_small_bytes ? need : (size_t)c->host_small_bytes;
                for (int i = 0; i < DXO_HOST_SLOTS; ++i) DXO_HIP(c, hipMalloc(&c->slot_buf[i], want));
                c->slot_bytes = want;
            }
            const bool timed = c->timing != 0;
            if (timed)
                for (auto& e : c->small_ev)
                    if (!e) DXO_HIP(c, hipEventCreate(&e));
            const auto t0s = std::chrono::steady_clock::now();
            hipStream_t s = c->slot_stream[0];
            char* hbase = static_cast<char*>(c->small_pinned);
            char* dbase = static_cast<char*>(c->slot_buf[0]);
            // ---- no DMA at all (option "host_zero_copy_bytes", 0 = off): the kernel reads and writes page-locked, device-mapped HOST
            // memory over PCIe directly — the caller's own arrays where they are page-locked blocks of this library, the context's
            // staging block otherwise; spans that live on the device (dxo_vm_state) are used where they are. What is left of a call
            // is the pack of the pageable inputs, one launch per piece and one stream wait.
            bool zero_copy = stream_once && !timed && (int64_t)need <= c->host_zero_copy_bytes;
            if (zero_copy) {
                void* mapped = nullptr;
                if (hipHostGetDevicePointer(&mapped, hbase, 0) != hipSuccess || !mapped) {
                    (void)hipGetLastError();
                    zero_copy = false;
                } else {
                    // Every span is either DIRECT — the caller's array is page-locked memory this library handed out or registered
                    // (the factories' output arrays are: Context.pinned_recycled), the kernel reads / writes it in place and no
                    // host copy is made — or STAGED in the context's pinned block (packed before the launch / unpacked after it).
                    // The batch runs in up to 4 PIECES on 64-unit borders: piece k + 1 is packed by this thread while the kernel
                    // of piece k moves its bytes over PCIe, so the pack of all but the first piece is hidden.
                    char* mbase = static_cast<char*>(mapped);
                    const size_t ni = inputs.size(), no = outputs.size();
                    std::vector<char*> z_in(ni), z_out(no), st_in(ni, nullptr), st_out(no, nullptr);
                    size_t zo = 0;
                    for (size_t k = 0; k < ni; ++k) {
                        const size_t bytes = inputs[k].bytes_pp * (size_t)n;
                        if (inputs[k].dev) {                  // lives on the device (dxo_vm_state): read where it is
                            z_in[k] = static_cast<char*>(inputs[k].dev);
                            continue;
                        }
                        if (void* m = dxo_pinned_mapped(inputs[k].in, bytes)) {
                            z_in[k] = static_cast<char*>(m);
                        } else {
                            st_in[k] = hbase + zo;
                            z_in[k] = mbase + zo;
                        }
                        zo += round_up(bytes, 256);
                    }
                    std::vector<char*> dev_to_host(no, nullptr);      // outputs the kernel writes on the DEVICE and the caller wants too
                    for (size_t k = 0; k < no; ++k) {
                        const size_t bytes = outputs[k].bytes_pp * (size_t)n;
                        void* m = outputs[k].out ? dxo_pinned_mapped(outputs[k].out, bytes) : nullptr;
                        if (outputs[k].dev) {                 // result stays in the device mirror; one D2H copy brings the caller's copy
                            z_out[k] = static_cast<char*>(outputs[k].dev);
                            if (outputs[k].out) {
                                if (m) dev_to_host[k] = static_cast<char*>(outputs[k].out);
                                else { st_out[k] = hbase + zo; dev_to_host[k] = hbase + zo; }
                            }
                        } else if (m) {
                            z_out[k] = static_cast<char*>(m);
                        } else {
                            st_out[k] = hbase + zo;
                            z_out[k] = mbase + zo;
                        }
                        zo += round_up(bytes, 256);
                    }
                    int64_t pieces = c->host_zero_copy_piece_bytes > 0 ? (int64_t)(need / (size_t)c->host_zero_copy_piece_bytes) : 1;
                    if (pieces > 4) pieces = 4;
                    if (pieces < 1) pieces = 1;
                    int64_t step = (n + pieces - 1) / pieces;
                    step = (step + DXO_WAVE - 1) / DXO_WAVE * DXO_WAVE;
                    std::vector<void*> a_in(ni), a_out(no);
                    for (int64_t b0 = 0; b0 < n; b0 += step) {
                        const int64_t m = b0 + step < n ? step : n - b0;
                        for (size_t k = 0; k < ni; ++k) {
                            const size_t o = inputs[k].bytes_pp * (size_t)b0;
                            if (st_in[k]) std::memcpy(st_in[k] + o, static_cast<const char*>(inputs[k].in) + o, inputs[k].bytes_pp * (size_t)m);
                            a_in[k] = z_in[k] + o;
                        }
                        for (size_t k = 0; k < no; ++k) a_out[k] = z_out[k] + outputs[k].bytes_pp * (size_t)b0;
                        const int rcz = launch(c, user, m, a_in.data(), a_out.data(), s);
                        if (rcz != DXO_OK) return rcz;
                    }
                    DXO_HIP(c, hipGetLastError());
                    for (size_t k = 0; k < no; ++k)
                        if (dev_to_host[k]) DXO_HIP(c, hipMemcpyAsync(dev_to_host[k], outputs[k].dev, outputs[k].bytes_pp * (size_t)n, hipMemcpyDeviceToHost, s));
                    DXO_HIP(c, hipStreamSynchronize(s));
                    for (size_t k = 0; k < no; ++k)
                        if (st_out[k] && outputs[k].out) std::memcpy(outputs[k].out, st_out[k], outputs[k].bytes_pp * (size_t)n);
                    if (post) {
                        const int rcz = post(c, user, 0, n);
                        if (rcz != DXO_OK) return rcz;
                    }
                    c->last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0s).count();
                    return DXO_OK;
                }
            }
            std::vector<void*> d_in(inputs.size()), d_out(outputs.size());
            size_t off = 0;
            for (size_t k = 0; k < inputs.size(); ++k) {
                if (inputs[k].dev) {
                    d_in[k] = inputs[k].dev;
                    continue;
                }
                std::memcpy(hbase + off, inputs[k].in, inputs[k].bytes_pp * (size_t)n);
                d_in[k] = dbase + off;
                off += round_up(inputs[k].bytes_pp * (size_t)n, 256);
            }
            if (timed) DXO_HIP(c, hipEventRecord(c->small_ev[0], s));
            if (in_bytes) DXO_HIP(c, hipMemcpyAsync(dbase, hbase, in_bytes, hipMemcpyHostToDevice, s));
            if (timed) DXO_HIP(c, hipEventRecord(c->small_ev[1], s));
            const size_t out_base = off;
            for (size_t k = 0; k < outputs.size(); ++k) {
                d_out[k] = outputs[k].dev ? outputs[k].dev : (void*)(dbase + off);
                off += round_up(outputs[k].bytes_pp * (size_t)n, 256);
            }
            int rc = launch(c, user, n, d_in.data(), d_out.data(), s);
            if (rc != DXO_OK) return rc;
            DXO_HIP(c, hipGetLastError());
            if (timed) DXO_HIP(c, hipEventRecord(c->small_ev[2], s));
            {   // device-resident outputs join the staging block so that ONE copy brings everything back
                size_t o2 = out_base;
                for (size_t k = 0; k < outputs.size(); ++k) {
                    if (outputs[k].dev && outputs[k].out)
                        DXO_HIP(c, hipMemcpyAsync(dbase + o2, outputs[k].dev, outputs[k].bytes_pp * (size_t)n, hipMemcpyDeviceToDevice, s));
                    o2 += round_up(outputs[k].bytes_pp * (size_t)n, 256);
                }
            }
            DXO_HIP(c, hipMemcpyAsync(hbase + out_base, dbase + out_base, out_bytes, hipMemcpyDeviceToHost, s));
            if (timed) DXO_HIP(c, hipEventRecord(c->small_ev[3], s));
            DXO_HIP(c, hipStreamSynchronize(s));
            off = out_base;
            for (size_t k = 0; k < outputs.size(); ++k) {
                if (outputs[k].out) std::memcpy(outputs[k].out, hbase + off, outputs[k].bytes_pp * (size_t)n);
                off += round_up(outputs[k].bytes_pp * (size_t)n, 256);
            }
            if (post) {
                rc = post(c, user, 0, n);
                if (rc != DXO_OK) return rc;
            }
            if (timed) {
                float a = 0, b = 0, d = 0;
                DXO_HIP(c, hipEventElapsedTime(&a, c->small_ev[0], c->small_ev[1]));
                DXO_HIP(c, hipEventElapsedTime(&b, c->small_ev[1], c->small_ev[2]));
                DXO_HIP(c, hipEventElapsedTime(&d, c->small_ev[2], c->small_ev[3]));
                c->last.h2d_ms = a; c->last.kernel_ms = b; c->last.d2h_ms = d;
            }
            c->last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0s).count();
            return DXO_OK;
        }
    }
    int64_t chunk = c->host_chunk_points / (points_per_unit > 0 ? points_per_unit : 1);
    if (chunk < DXO_WAVE) chunk = DXO_WAVE;
    if (chunk > n) chunk = n;
    if (chunk < n) chunk = chunk / DXO_WAVE * DXO_WAVE;  // interior chunk borders on whole wave tiles
    // slot layout: every span starts on a 256-byte border
    size_t need = 0;
    for (const auto& s : inputs)
        if (!s.dev) need += round_up(s.bytes_pp * (size_t)chunk, 256);
    for (const auto& s : outputs)
        if (!s.dev) need += round_up(s.bytes_pp * (size_t)chunk, 256);
    if (need < 256) need = 256;
    if (need > c->slot_bytes) {
        for (int i = 0; i < DXO_HOST_SLOTS; ++i) {
            DXO_HIP(c, hipStreamSynchronize(c->slot_stream[i]));
            if (c->slot_buf[i]) DXO_HIP(c, hipFree(c->slot_buf[i]));
            c->slot_buf[i] = nullptr;
        }
        c->slot_bytes = 0;
        for (int i = 0; i < DXO_HOST_SLOTS; ++i) DXO_HIP(c, hipMalloc(&c->slot_buf[i], need));
        c->slot_bytes = need;
    }
    SlotEvents ev[DXO_HOST_SLOTS];
    int rc = DXO_OK;
    // Host stage (only with a `post` hook): its own thread waits for a chunk's D2H copies and runs the hook (which
    // fans out over the context's worker threads) while the calling thread keeps enqueueing the next chunks — the host
    // half of chunk i overlaps the PCIe traffic of chunks i+1.. instead of sitting between two enqueues.
    struct StageItem { int slot; int64_t first, m; };
    hipEvent_t post_ev[DXO_HOST_SLOTS] = {nullptr, nullptr, nullptr};
    bool post_pending[DXO_HOST_SLOTS] = {false, false, false};
    std::mutex st_m;
    std::condition_variable st_cv;
    std::deque<StageItem> st_q;
    bool st_done = false;
    int st_rc = DXO_OK;
    std::thread stage;
    auto destroy_events = [&]() {
        for (auto& s : ev)
            for (auto& e : s.e)
                if (e) (void)hipEventDestroy(e);
        for (auto& e : post_ev)
            if (e) (void)hipEventDestroy(e);
    };
    if (post) {
        for (auto& e : post_ev) {
            hipError_t err = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (err != hipSuccess) {
                destroy_events();
                return dxo_hip_fail(c, err, "hipEventCreate");
            }
        }
        stage = std::thread([&]() {
            (void)hipSetDevice(c->device);
            for (;;) {
                StageItem it;
                {
                    std::unique_lock<std::mutex> lk(st_m);
                    st_cv.wait(lk, [&] { return st_done || !st_q.empty(); });
                    if (st_q.empty()) return;
                    it = st_q.front();
                    st_q.pop_front();
                }
                const hipError_t e = hipEventSynchronize(post_ev[it.slot]);
                {
                    std::lock_guard<std::mutex> lk(st_m);
                    post_pending[it.slot] = false;   // the slot's event may be recorded again
                    if (e != hipSuccess && st_rc == DXO_OK) st_rc = (int)e > 0 ? (int)e : 999;
                }
                st_cv.notify_all();
                if (e == hipSuccess) {
                    const int r = post(c, user, it.first, it.m);
                    if (r != DXO_OK) {
                        std::lock_guard<std::mutex> lk(st_m);
                        if (st_rc == DXO_OK) st_rc = r;
                    }
                }
            }
        });
    }
    auto stop_stage = [&]() {
        if (!stage.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(st_m);
            st_done = true;
        }
        st_cv.notify_all();
        stage.join();
    };
    for (auto& s : ev)
        for (auto& e : s.e) {
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) {
                stop_stage();
                destroy_events();
                return dxo_hip_fail(c, err, "hipEventCreate");
            }
        }
    auto harvest = [&](int slot) -> int {
        if (!ev[slot].used) return DXO_OK;
        DXO_HIP(c, hipEventSynchronize(ev[slot].e[3]));
        float a = 0, b = 0, d = 0;
        DXO_HIP(c, hipEventElapsedTime(&a, ev[slot].e[0], ev[slot].e[1]));
        DXO_HIP(c, hipEventElapsedTime(&b, ev[slot].e[1], ev[slot].e[2]));
        DXO_HIP(c, hipEventElapsedTime(&d, ev[slot].e[2], ev[slot].e[3]));
        c->last.h2d_ms += a;
        c->last.kernel_ms += b;
        c->last.d2h_ms += d;
        ev[slot].used = false;
        if (post) {   // the stage must have seen this slot's completion event before the event is recorded again
            std::unique_lock<std::mutex> lk(st_m);
            st_cv.wait(lk, [&] { return !post_pending[slot]; });
        }
        return DXO_OK;
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<void*> d_in(inputs.size()), d_out(outputs.size());
    int64_t done = 0;
    for (int64_t i = 0; done < n && rc == DXO_OK; ++i) {
        const int slot = (int)(i % DXO_HOST_SLOTS);
        const int64_t m = (n - done < chunk) ? (n - done) : chunk;
        hipStream_t s = c->slot_stream[slot];
        if ((rc = harvest(slot)) != DXO_OK) break;
        char* base = (char*)c->slot_buf[slot];
        size_t off = 0;
        hipError_t e = hipEventRecord(ev[slot].e[0], s);
        for (size_t k = 0; k < inputs.size() && e == hipSuccess; ++k) {
            if (inputs[k].dev) {
                d_in[k] = (char*)inputs[k].dev + (size_t)done * inputs[k].bytes_pp;
                continue;
            }
            d_in[k] = base + off;
            off += round_up(inputs[k].bytes_pp * (size_t)chunk, 256);
            e = hipMemcpyAsync(d_in[k], (const char*)inputs[k].in + (size_t)done * inputs[k].bytes_pp,
                               inputs[k].bytes_pp * (size_t)m, hipMemcpyHostToDevice, s);
        }
        if (e == hipSuccess) e = hipEventRecord(ev[slot].e[1], s);
        if (e != hipSuccess) { rc = dxo_hip_fail(c, e, "host pipeline H2D"); break; }
        for (size_t k = 0; k < outputs.size(); ++k) {
            if (outputs[k].dev) {
                d_out[k] = (char*)outputs[k].dev + (size_t)done * outputs[k].bytes_pp;
                continue;
            }
            d_out[k] = base + off;
            off += round_up(outputs[k].bytes_pp * (size_t)chunk, 256);
        }
        rc = launch(c, user, m, d_in.data(), d_out.data(), s);
        if (rc != DXO_OK) break;
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(ev[slot].e[2], s);
        for (size_t k = 0; k < outputs.size() && e == hipSuccess; ++k) {
            if (!outputs[k].out) continue;
            e = hipMemcpyAsync((char*)outputs[k].out + (size_t)done * outputs[k].bytes_pp, d_out[k],
                               outputs[k].bytes_pp * (size_t)m, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipEventRecord(ev[slot].e[3], s);
        if (e == hipSuccess && post) e = hipEventRecord(post_ev[slot], s);
        if (e != hipSuccess) { rc = dxo_hip_fail(c, e, "host pipeline D2H"); break; }
        ev[slot].used = true;
        if (post) {
            {
                std::lock_guard<std::mutex> lk(st_m);
                post_pending[slot] = true;
                st_q.push_back({slot, done, m});
            }
            st_cv.notify_all();
        }
        done += m;
    }
    for (int slot = 0; slot < DXO_HOST_SLOTS; ++slot) {
        int r2 = harvest(slot);
        if (rc == DXO_OK) rc = r2;
        hipError_t e = hipStreamSynchronize(c->slot_stream[slot]);
        if (rc == DXO_OK && e != hipSuccess) rc = dxo_hip_fail(c, e, "host pipeline sync");
    }
    stop_stage();   // drains the queue: every chunk's host half has run when this returns
    if (rc == DXO_OK && st_rc != DXO_OK) rc = st_rc;
    destroy_events();
    c->last.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
