// mgpu.hip — dxo_mgpu_*: cell-block sharding over the GPUs of one node with the RCCL all-gather INSIDE the library
// (BASELINE north_star; SURVEY.md 8b "dxo_mgpu_* variants taking a device list + doing the all-gather", 8e).
//
// The reference never gathers quadrature data: every MPI rank evaluates its own mesh partition
// (src/dolfinx_external_operator/external_operator.py:365-371) and halo-updates the coefficient (:445). Here the
// quadrature points of ONE coefficient vector are split into contiguous cell blocks (arrays are cell-major, so a
// block is a contiguous slice of every array), each GPU runs the pointwise kernel on its block and writes the result
// straight into its slice of a FULL-length output array; RCCL's in-place all-gather (sendbuff == recvbuff +
// rank * count) then gives every GPU the whole vector — nothing is copied locally.
//
// Two forms of one object:
//   dxo_mgpu_create       single process, n_dev devices: one dxo_ctx + one communicator per device (ncclCommInitAll),
//                         collectives issued for all local devices inside one ncclGroupStart/End.
//   dxo_mgpu_create_rank  one process per GPU (MPI ranks of a DOLFINx run, torch.distributed workers): the caller
//                         broadcasts the 128-byte id of dxo_mgpu_unique_id, every rank joins with ncclCommInitRank.
// Gather modes: FULL = all-gather of (C_tang, sigma, dp), (d*d+d+1) doubles per point over xGMI; COMPACT = all-gather
// of (sigma, dp) only ((d+1) doubles, 6.1x fewer link bytes at d = 6) + rebuild of EVERY block's tangent — the rank's
// own included — from the gathered state on the device (dxo_vm_expand_tangent, HBM-bound): xGMI (7 links x ~153 GB/s
// per GPU), not HBM, is the roof of the reassembly. Because every rank runs the same rebuild on the same gathered
// (sigma, dp), the replicas of the coefficient vector are BIT-IDENTICAL across ranks, and the reference's 0/0 point
// (f_elastic == 0, demo_plasticity_von_mises.py:318) travels as the sign bit of dp and comes out as the reference's NaN
// tangent on every rank (the marks are cleared afterwards: dp is +0 there, as in the reference).
//
// Buffers handed to RCCL must be ordinary hipMalloc memory: the group's contexts get "placement_vmm" = 0, and the
// collectives refuse (DXO_E_MEM) a pointer inside an arena block built from 2 MB physical chunks (accessible from its
// own device only, not exportable through hipIpcGetMemHandle).
//
// RCCL is resolved lazily (dlopen "librccl.so.1" at the first dxo_mgpu_* call): libdxo_hip.so has no link-time
// dependency on it, single-GPU users never load it, and inside a PyTorch process the copy PyTorch has already
// loaded is the one that is used (same SONAME), so there is ONE RCCL per process.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include <thread>

#include "host_pool.h"

#include "dxo_common.h"

struct dxo_mgpu {
    int world = 0;                 // ranks in the communicator
    std::vector<int> rank;         // global rank of each LOCAL device
    std::vector<dxo_ctx*> ctx;     // one per local device
    std::vector<bool> own_ctx;
    std::vector<ncclComm_t> comm;
    bool local_only = false;       // dxo_mgpu_create_local: contexts without a communicator (host-sharded calls only)
    // DXO_GATHER_COMPACT_PIPELINED: the exchange runs on its own stream per local device, beside the kernels of later chunks
    std::vector<hipStream_t> xstream;
    std::vector<hipEvent_t> ev_kernel, ev_arrived;      // [local device][chunk], made on first use
    int ev_chunks = 0;
    std::string err;
};

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool is_mock = false;      // the library loaded is tests/mock_rccl (it exports mock_rccl_bytes_moved), not RCCL
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.why = std::string("RCCL not found (dlopen librccl.so.1): ") + (e ? e : "");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p && r.why.empty()) r.why = std::string("RCCL lacks symbol ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.is_mock = dlsym(r.handle, "mock_rccl_bytes_moved") != nullptr;
    });
    return r.why.empty() ? &r : nullptr;
}

int mg_fail(dxo_mgpu* g, int code, const std::string& what) {
    if (g) g->err = what;
    return code;
}

// RCCL errors are reported as positive codes above the HIP range so they cannot be mistaken for a hipError_t
int nccl_fail(dxo_mgpu* g, ncclResult_t r, const char* where) {
    Rccl* R = rccl();
    if (g) g->err = std::string(where) + ": " + (R && R->GetErrorString ? R->GetErrorString(r) : "RCCL error") + " (ncclResult " + std::to_string((int)r) + ")";
    return 10000 + (int)r;
}

#define DXO_NCCL(g, call)                                             \
    do {                                                              \
        ncclResult_t r_ = (call);                                     \
        if (r_ != ncclSuccess) return nccl_fail((g), r_, #call);      \
    } while (0)

// HIP errors are positive hipError_t codes (as everywhere in the C ABI)
int mg_hip_fail(dxo_mgpu* g, hipError_t e, const char* what) {
    (void)hipGetLastError();      // clear the sticky copy; `e` is the failing call's own return value
    if (g) g->err = std::string(what) + ": " + hipGetErrorString(e) + " (hipError " + std::to_string((int)e) + ")";
    return (int)e > 0 ? (int)e : 999;
}

int need_rccl(dxo_mgpu* g) {
    if (g->local_only)
        return mg_fail(g, DXO_E_OPTION, "this group was made by dxo_mgpu_create_local: it has no communicator (host-sharded calls only)");
    if (rccl()) return DXO_OK;
    return mg_fail(g, DXO_E_NODEVICE, "RCCL is not available in this process (librccl.so.1 could not be loaded)");
}

}  // namespace

extern "C" {

const char* dxo_mgpu_last_error(const dxo_mgpu* g) { return g ? g->err.c_str() : "null dxo_mgpu"; }

int dxo_mgpu_size(const dxo_mgpu* g) { return g ? g->world : DXO_E_NULL; }

int dxo_mgpu_local_count(const dxo_mgpu* g) { return g ? (int)g->ctx.size() : DXO_E_NULL; }

int dxo_mgpu_rank(const dxo_mgpu* g, int i) {
    if (!g) return DXO_E_NULL;
    if (i < 0 || i >= (int)g->rank.size()) return DXO_E_SIZE;
    return g->rank[(size_t)i];
}

dxo_ctx* dxo_mgpu_ctx(dxo_mgpu* g, int i) {
    if (!g || i < 0 || i >= (int)g->ctx.size()) return nullptr;
    return g->ctx[(size_t)i];
}

int dxo_mgpu_destroy(dxo_mgpu* g) {
    if (!g) return DXO_E_NULL;
    Rccl* R = g->local_only ? nullptr : rccl();   // a local group never loads RCCL
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (g->ctx[i]) {
            (void)hipSetDevice(g->ctx[i]->device);
            (void)hipStreamSynchronize(dxo_launch_stream(g->ctx[i]));
        }
        if (R && i < g->comm.size() && g->comm[i]) (void)R->CommDestroy(g->comm[i]);
        if (i < g->xstream.size() && g->xstream[i]) (void)hipStreamDestroy(g->xstream[i]);
    }
    for (hipEvent_t e : g->ev_kernel) (void)hipEventDestroy(e);
    for (hipEvent_t e : g->ev_arrived) (void)hipEventDestroy(e);
    for (size_t i = 0; i < g->ctx.size(); ++i)
        if (g->own_ctx[i] && g->ctx[i]) (void)dxo_ctx_destroy(g->ctx[i]);
    delete g;
    return DXO_OK;
}

int dxo_mgpu_create(const int* devices, int n_dev, dxo_mgpu** out) {
    if (!out) return DXO_E_NULL;
    *out = nullptr;
    if (n_dev < 1) return DXO_E_SIZE;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return DXO_E_NODEVICE;
    }
    std::vector<int> devs((size_t)n_dev);
    for (int i = 0; i < n_dev; ++i) {
        devs[(size_t)i] = devices ? devices[i] : i;
        if (devs[(size_t)i] < 0 || devs[(size_t)i] >= count) return DXO_E_NODEVICE;
    }
    if (!rccl()) return DXO_E_NODEVICE;
    // one rank per physical device (RCCL refuses anything else). TEST HOOK: over the MOCK transport only (tests/mock_rccl, recognised by
    // its marker symbol — never over RCCL, where a shared device would end in an ncclCommInit failure or a hang) and with
    // DXO_MGPU_TEST_SHARE_DEVICE=1 in the environment, several ranks may share a device: the N > 1 data path on a one-GPU box.
    const char* share = std::getenv("DXO_MGPU_TEST_SHARE_DEVICE");
    const bool may_share = rccl()->is_mock && share && share[0] == '1';
    for (int i = 0; i < n_dev; ++i) {
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)j] == devs[(size_t)i] && !may_share) return DXO_E_SIZE;
    }
    dxo_mgpu* g = new dxo_mgpu();
    g->world = n_dev;
    g->comm.assign((size_t)n_dev, nullptr);
    for (int i = 0; i < n_dev; ++i) {
        dxo_ctx* c = nullptr;
        const int rc = dxo_ctx_create(devs[(size_t)i], &c);
        if (rc != DXO_OK) {
            dxo_mgpu_destroy(g);
            return rc;
        }
        c->placement_vmm = 0;   // arena blocks of this context may become RCCL buffers: hipMalloc candidates only
        g->ctx.push_back(c);
        g->own_ctx.push_back(true);
        g->rank.push_back(i);
    }
    const ncclResult_t r = rccl()->CommInitAll(g->comm.data(), n_dev, devs.data());
    if (r != ncclSuccess) {
        const int code = nccl_fail(g, r, "ncclCommInitAll");
        std::fprintf(stderr, "dxo_mgpu_create: %s\n", g->err.c_str());
        for (auto& cm : g->comm) cm = nullptr;
        dxo_mgpu_destroy(g);
        return code;
    }
    *out = g;
    return DXO_OK;
}

// Contexts only, no communicator: for calls whose data path has no exchange step (dxo_mgpu_von_mises_host). RCCL is
// not loaded. A device may appear more than once (two pipelines on one GPU: the single-GPU test of the sharded path).
int dxo_mgpu_create_local(const int* devices, int n_dev, dxo_mgpu** out) {
    if (!out) return DXO_E_NULL;
    *out = nullptr;
    if (n_dev < 1) return DXO_E_SIZE;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return DXO_E_NODEVICE;
    }
    dxo_mgpu* g = new dxo_mgpu();
    g->world = n_dev;
    g->local_only = true;
    g->comm.assign((size_t)n_dev, nullptr);
    for (int i = 0; i < n_dev; ++i) {
        const int dev = devices ? devices[i] : i;
        dxo_ctx* c = nullptr;
        const int rc = dev < 0 || dev >= count ? DXO_E_NODEVICE : dxo_ctx_create(dev, &c);
        if (rc != DXO_OK) {
            dxo_mgpu_destroy(g);
            return rc;
        }
        g->ctx.push_back(c);
        g->own_ctx.push_back(true);
        g->rank.push_back(i);
    }
    *out = g;
    return DXO_OK;
}

// Host arrays of ALL n points, split into one contiguous block per local device (borders on 64-point tiles); every
// device runs dxo_von_mises(DXO_MEM_HOST) on its block from its own thread — its own chunked H2D / kernel / D2H pipeline
// over its own PCIe link — and the results land in the caller's arrays directly: no exchange step, no collective.
// This is how a single-process caller with NumPy arrays uses N GPUs: the host path is PCIe-bound, so N
// links are what scales it. Context options (vm_host_tangent, host_chunk_points ...) are those of each local context
// (dxo_mgpu_ctx); the contexts' host worker threads share the process's CPU budget.
int dxo_mgpu_von_mises_host(dxo_mgpu* g, const dxo_vm_params* prm, int d, int64_t n, const double* deps, const double* sigma_n,
                            const double* p, double* C_tang, double* sigma, double* dp) {
    if (!g) return DXO_E_NULL;
    if (!prm) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_von_mises_host: params is NULL");
    if (d != 4 && d != 6) return mg_fail(g, DXO_E_DIM, "dxo_mgpu_von_mises_host: d must be 4 or 6");
    if (n < 0) return mg_fail(g, DXO_E_SIZE, "dxo_mgpu_von_mises_host: n < 0");
    if (n > 0 && (!deps || !sigma_n || !p || !C_tang || !sigma || !dp)) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_von_mises_host: NULL array");
    if (n == 0) return DXO_OK;
    const int64_t L = (int64_t)g->ctx.size();
    int64_t block = (n + L - 1) / L;
    block = (block + DXO_WAVE - 1) / DXO_WAVE * DXO_WAVE;
    // the workers of all contexts together stay inside the CPU budget of the process (host_pool.h)
    const int share = (int)((dxo_host_cpu_budget() - 2) / L);
    std::vector<int64_t> saved((size_t)L, 0);
    for (int64_t i = 0; i < L; ++i) {
        (void)dxo_ctx_get_option(g->ctx[(size_t)i], "host_threads", &saved[(size_t)i]);
        if (saved[(size_t)i] > share) (void)dxo_ctx_set_option(g->ctx[(size_t)i], "host_threads", share > 1 ? share : 1);
    }
    std::vector<int> rc((size_t)L, DXO_OK);
    std::vector<std::thread> th;
    for (int64_t i = 0; i < L; ++i) {
        const int64_t b = i * block, e = b + block < n ? b + block : n;
        if (b >= e) break;
        th.emplace_back([=, &rc] {
            rc[(size_t)i] = dxo_von_mises(g->ctx[(size_t)i], prm, d, e - b, DXO_MEM_HOST, deps + b * d, sigma_n + b * d, p + b,
                                          C_tang + b * d * d, sigma + b * d, dp + b);
        });
    }
    for (auto& t : th) t.join();
    for (int64_t i = 0; i < L; ++i) (void)dxo_ctx_set_option(g->ctx[(size_t)i], "host_threads", saved[(size_t)i]);
    for (int64_t i = 0; i < L; ++i)
        if (rc[(size_t)i] != DXO_OK) return mg_fail(g, rc[(size_t)i], dxo_last_error(g->ctx[(size_t)i]));
    return DXO_OK;
}

int dxo_mgpu_unique_id(void* id128) {
    if (!id128) return DXO_E_NULL;
    if (!rccl()) return DXO_E_NODEVICE;
    static_assert(sizeof(ncclUniqueId) == DXO_MGPU_ID_BYTES, "dxo.h promises a 128-byte id");
    ncclUniqueId id;
    const ncclResult_t r = rccl()->GetUniqueId(&id);
    if (r != ncclSuccess) return 10000 + (int)r;
    std::memcpy(id128, &id, sizeof id);
    return DXO_OK;
}

int dxo_mgpu_create_rank(dxo_ctx* ctx, const void* id128, int rank, int world, dxo_mgpu** out) {
    if (!ctx || !id128 || !out) return DXO_E_NULL;
    DXO_LOCK(ctx);
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mgpu_create_rank: need 0 <= rank < world");
    if (!rccl()) return dxo_fail(ctx, DXO_E_NODEVICE, "dxo_mgpu_create_rank: RCCL (librccl.so.1) could not be loaded");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    dxo_mgpu* g = new dxo_mgpu();
    g->world = world;
    ctx->placement_vmm = 0;   // from here on this context's arena hands out hipMalloc blocks only (RCCL buffers)
    g->ctx.push_back(ctx);
    g->own_ctx.push_back(false);
    g->rank.push_back(rank);
    g->comm.assign(1, nullptr);
    const ncclResult_t r = rccl()->CommInitRank(&g->comm[0], world, id, rank);
    if (r != ncclSuccess) {
        const int code = nccl_fail(g, r, "ncclCommInitRank");
        dxo_fail(ctx, code, g->err.c_str());
        g->comm[0] = nullptr;
        dxo_mgpu_destroy(g);
        return code;
    }
    *out = g;
    return DXO_OK;
}

int dxo_mgpu_synchronize(dxo_mgpu* g) {
    if (!g) return DXO_E_NULL;
    for (auto* c : g->ctx) {
        const int rc = dxo_ctx_synchronize(c);
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(c));
    }
    return DXO_OK;
}

// In-place all-gather of `bytes_per_rank` bytes per rank on every local device: buf[i] is the FULL-length array of local
// device i, whose own block already sits at offset rank * bytes_per_rank.
static int mg_all_gather_bytes(dxo_mgpu* g, void* const* buf, size_t bytes_per_rank) {
    const int rc0 = need_rccl(g);
    if (rc0 != DXO_OK) return rc0;
    if (bytes_per_rank == 0) return DXO_OK;
    Rccl* R = rccl();
    DXO_NCCL(g, R->GroupStart());
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (!buf[i]) {
            (void)R->GroupEnd();
            return mg_fail(g, DXO_E_NULL, "dxo_mgpu_all_gather: NULL buffer");
        }
        if (dxo_arena_is_vmm(g->ctx[i], buf[i])) {
            (void)R->GroupEnd();
            return mg_fail(g, DXO_E_MEM, "dxo_mgpu_all_gather: the buffer lies in an arena block backed by 2 MB physical chunks, which peers "
                                         "cannot access; allocate it with option placement_vmm = 0 (the group's contexts have it set)");
        }
        char* b = static_cast<char*>(buf[i]);
        const ncclResult_t r = R->AllGather(b + (size_t)g->rank[i] * bytes_per_rank, b, bytes_per_rank, ncclChar, g->comm[i],
                                            dxo_launch_stream(g->ctx[i]));
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            return nccl_fail(g, r, "ncclAllGather");
        }
    }
    DXO_NCCL(g, R->GroupEnd());
    return DXO_OK;
}

// The same result as the in-place all-gather, as point-to-point traffic (SURVEY.md 8e prefers it on a fully connected xGMI
// node: one link per GPU pair, so every block travels on its own link at once whatever algorithm RCCL's all-gather would pick for
// the message size). Piece [off, off + len) ELEMENTS of every rank's block (block = count_per_rank elements of elem_bytes bytes;
// rank r's block at element r * count_per_rank of the full-length array): each local device sends its own piece to every peer
// and receives each peer's piece straight into place; all 2 (world - 1) operations of all local devices in ONE group. Peer order
// staggered by rank (at step s everybody talks to rank + s / rank - s: disjoint pairs).
static int mg_exchange_direct(dxo_mgpu* g, void* const* buf, size_t elem_bytes, size_t count_per_rank, size_t off, size_t len,
                              const std::vector<hipStream_t>* streams) {
    const int rc0 = need_rccl(g);
    if (rc0 != DXO_OK) return rc0;
    if (len == 0 || g->world == 1) return DXO_OK;
    Rccl* R = rccl();
    if (!R->Send || !R->Recv) return mg_fail(g, DXO_E_NODEVICE, "this RCCL has no ncclSend / ncclRecv");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (!buf[i]) return mg_fail(g, DXO_E_NULL, "dxo_mgpu exchange: NULL buffer");
        if (dxo_arena_is_vmm(g->ctx[i], buf[i]))
            return mg_fail(g, DXO_E_MEM, "dxo_mgpu exchange: the buffer lies in an arena block backed by 2 MB physical chunks, which peers cannot "
                                         "access; allocate it with option placement_vmm = 0 (the group's contexts have it set)");
    }
    DXO_NCCL(g, R->GroupStart());
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        char* b = static_cast<char*>(buf[i]);
        const int me = g->rank[i];
        hipStream_t st = streams ? (*streams)[i] : dxo_launch_stream(g->ctx[i]);
        const char* mine = b + ((size_t)me * count_per_rank + off) * elem_bytes;
        for (int step = 1; step < g->world; ++step) {
            const int to = (me + step) % g->world, from = (me - step + g->world) % g->world;
            ncclResult_t r = R->Send(mine, len * elem_bytes, ncclChar, to, g->comm[i], st);
            if (r == ncclSuccess) r = R->Recv(b + ((size_t)from * count_per_rank + off) * elem_bytes, len * elem_bytes, ncclChar, from, g->comm[i], st);
            if (r != ncclSuccess) {
                (void)R->GroupEnd();
                return nccl_fail(g, r, "ncclSend / ncclRecv");
            }
        }
    }
    DXO_NCCL(g, R->GroupEnd());
    return DXO_OK;
}

// the exchange streams and the per-chunk events of DXO_GATHER_COMPACT_PIPELINED, made once per group
static int mg_pipeline_resources(dxo_mgpu* g, int chunks) {
    const size_t L = g->ctx.size();
    if (g->xstream.size() != L) {
        g->xstream.assign(L, nullptr);
        for (size_t i = 0; i < L; ++i) {
            hipError_t e = hipSetDevice(g->ctx[i]->device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->xstream[i], hipStreamNonBlocking);
            if (e != hipSuccess) {
                // leave nothing half-made: a later call must create the streams again, never run the exchange on a null (legacy default) stream
                for (size_t j = 0; j < i; ++j)
                    if (g->xstream[j]) (void)hipStreamDestroy(g->xstream[j]);
                g->xstream.clear();
                return mg_hip_fail(g, e, "dxo_mgpu: could not create the exchange stream");
            }
        }
    }
    if (g->ev_chunks < chunks) {
        for (hipEvent_t e : g->ev_kernel) (void)hipEventDestroy(e);
        for (hipEvent_t e : g->ev_arrived) (void)hipEventDestroy(e);
        g->ev_kernel.assign(L * (size_t)chunks, nullptr);
        g->ev_arrived.assign(L * (size_t)chunks, nullptr);
        for (size_t i = 0; i < L; ++i) {
            (void)hipSetDevice(g->ctx[i]->device);
            for (int k = 0; k < chunks; ++k) {
                hipError_t e = hipEventCreateWithFlags(&g->ev_kernel[i * chunks + k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_arrived[i * chunks + k], hipEventDisableTiming);
                if (e != hipSuccess) {
                    for (hipEvent_t ev : g->ev_kernel) if (ev) (void)hipEventDestroy(ev);
                    for (hipEvent_t ev : g->ev_arrived) if (ev) (void)hipEventDestroy(ev);
                    g->ev_kernel.clear();
                    g->ev_arrived.clear();
                    g->ev_chunks = 0;
                    return mg_hip_fail(g, e, "dxo_mgpu: could not create the pipeline events");
                }
            }
        }
        g->ev_chunks = chunks;
    }
    return DXO_OK;
}

int dxo_mgpu_all_gather(dxo_mgpu* g, double* const* buf, int64_t count_per_rank) {
    if (!g || !buf) return DXO_E_NULL;
    if (count_per_rank < 0) return mg_fail(g, DXO_E_SIZE, "dxo_mgpu_all_gather: negative count");
    return mg_all_gather_bytes(g, reinterpret_cast<void* const*>(buf), (size_t)count_per_rank * sizeof(double));
}

// ---- the other pointwise operators, sharded the same way: every local device evaluates its cell block into its slice of
// the FULL-length outputs, then one in-place all-gather per output array (DXO_GATHER_FULL) or none (DXO_GATHER_NONE,
// block-length outputs). There is no compact form for them (no output is a function of the others): DXO_GATHER_COMPACT
// is refused with DXO_E_OPTION. A NULL pointer ARRAY means "this output is not requested" (on every device).
}  // extern "C"

namespace {

struct MgOut {
    void* const* arr;       // per local device: full-length (gather) or block-length array; may be NULL = skip
    size_t bytes_pp;        // bytes per point
};

int mg_check(dxo_mgpu* g, int64_t n_per_rank, int gather, const char* who) {
    if (n_per_rank < 0) return mg_fail(g, DXO_E_SIZE, std::string(who) + ": n_per_rank < 0");
    if (gather == DXO_GATHER_COMPACT) return mg_fail(g, DXO_E_OPTION, std::string(who) + ": DXO_GATHER_COMPACT exists for von Mises only (its tangent is a function of (sigma, dp))");
    if (gather != DXO_GATHER_NONE && gather != DXO_GATHER_FULL) return mg_fail(g, DXO_E_OPTION, std::string(who) + ": bad gather mode");
    if (gather != DXO_GATHER_NONE && (n_per_rank % 4)) return mg_fail(g, DXO_E_ALIGN, std::string(who) + ": with a gather n_per_rank must be a multiple of 4 (16-byte aligned blocks of every array)");
    return DXO_OK;
}

// slice of local device i inside output k (NULL stays NULL)
template <class T>
T* mg_slice(const dxo_mgpu* g, const MgOut& o, size_t i, int gather, size_t n) {
    if (!o.arr || !o.arr[i]) return nullptr;
    const size_t off = gather == DXO_GATHER_NONE ? 0 : (size_t)g->rank[i] * n * o.bytes_pp;
    return reinterpret_cast<T*>(static_cast<char*>(o.arr[i]) + off);
}

int mg_gather_outputs(dxo_mgpu* g, int gather, size_t n, std::initializer_list<MgOut> outs) {
    if (gather == DXO_GATHER_NONE || g->world == 1 || n == 0) return DXO_OK;
    for (const MgOut& o : outs) {
        if (!o.arr) continue;
        const int rc = mg_all_gather_bytes(g, o.arr, n * o.bytes_pp);
        if (rc != DXO_OK) return rc;
    }
    return DXO_OK;
}

}  // namespace

extern "C" {

int dxo_mgpu_mohr_coulomb(dxo_mgpu* g, const dxo_mc_params* prm, int64_t n_per_rank, int gather, const double* const* deps,
                          const double* const* sigma_n, double* const* C_tang, double* const* sigma, int32_t* const* niter,
                          double* const* yielding, double* const* norm_res, double* const* dlambda) {
    if (!g) return DXO_E_NULL;
    if (!prm || !deps || !sigma_n || !C_tang || !sigma) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_mohr_coulomb: NULL argument");
    int rc = mg_check(g, n_per_rank, gather, "dxo_mgpu_mohr_coulomb");
    if (rc != DXO_OK) return rc;
    const size_t n = (size_t)n_per_rank, sd = sizeof(double);
    const MgOut oC{(void* const*)C_tang, 16 * sd}, oS{(void* const*)sigma, 4 * sd}, oI{(void* const*)niter, sizeof(int32_t)},
        oY{(void* const*)yielding, sd}, oR{(void* const*)norm_res, sd}, oL{(void* const*)dlambda, sd};
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rc = dxo_mohr_coulomb(g->ctx[i], prm, n_per_rank, DXO_MEM_DEVICE, deps[i], sigma_n[i], mg_slice<double>(g, oC, i, gather, n),
                              mg_slice<double>(g, oS, i, gather, n), mg_slice<int32_t>(g, oI, i, gather, n), mg_slice<double>(g, oY, i, gather, n),
                              mg_slice<double>(g, oR, i, gather, n), mg_slice<double>(g, oL, i, gather, n));
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
    }
    return mg_gather_outputs(g, gather, n, {oC, oS, oI, oY, oR, oL});
}

int dxo_mgpu_icnn(dxo_mgpu* g, dxo_icnn* const* models, int precision, int64_t n_per_rank, int gather, const double* const* F,
                  double* const* dP, double* const* P) {
    if (!g) return DXO_E_NULL;
    if (!models || !F || !dP || !P) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_icnn: NULL argument");
    int rc = mg_check(g, n_per_rank, gather, "dxo_mgpu_icnn");
    if (rc != DXO_OK) return rc;
    const size_t n = (size_t)n_per_rank, sd = sizeof(double);
    const MgOut oD{(void* const*)dP, 16 * sd}, oP{(void* const*)P, 4 * sd};
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rc = dxo_icnn_eval(g->ctx[i], models[i], precision, n_per_rank, DXO_MEM_DEVICE, F[i], mg_slice<double>(g, oD, i, gather, n),
                           mg_slice<double>(g, oP, i, gather, n));
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
    }
    return mg_gather_outputs(g, gather, n, {oD, oP});
}

int dxo_mgpu_isihara(dxo_mgpu* g, const dxo_isihara_params* prm, int64_t n_per_rank, int gather, const double* const* F,
                     double* const* dP, double* const* P) {
    if (!g) return DXO_E_NULL;
    if (!prm || !F || !dP || !P) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_isihara: NULL argument");
    int rc = mg_check(g, n_per_rank, gather, "dxo_mgpu_isihara");
    if (rc != DXO_OK) return rc;
    const size_t n = (size_t)n_per_rank, sd = sizeof(double);
    const MgOut oD{(void* const*)dP, 16 * sd}, oP{(void* const*)P, 4 * sd};
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rc = dxo_isihara(g->ctx[i], prm, n_per_rank, DXO_MEM_DEVICE, F[i], mg_slice<double>(g, oD, i, gather, n), mg_slice<double>(g, oP, i, gather, n));
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
    }
    return mg_gather_outputs(g, gather, n, {oD, oP});
}

int dxo_mgpu_heat(dxo_mgpu* g, double A, double B, int gdim, int64_t n_per_rank, int gather, const double* const* T,
                  const double* const* sigma, double* const* q, double* const* dqdT, double* const* dqdsigma) {
    if (!g) return DXO_E_NULL;
    if (!T || !sigma) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_heat: NULL argument");
    if (gdim < 1 || gdim > 3) return mg_fail(g, DXO_E_DIM, "dxo_mgpu_heat: gdim must be 1, 2 or 3");
    int rc = mg_check(g, n_per_rank, gather, "dxo_mgpu_heat");
    if (rc != DXO_OK) return rc;
    const size_t n = (size_t)n_per_rank, sd = sizeof(double);
    const MgOut oQ{(void* const*)q, gdim * sd}, oT{(void* const*)dqdT, gdim * sd}, oS{(void* const*)dqdsigma, (size_t)gdim * gdim * sd};
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rc = dxo_heat(g->ctx[i], A, B, gdim, n_per_rank, DXO_MEM_DEVICE, T[i], sigma[i], mg_slice<double>(g, oQ, i, gather, n),
                      mg_slice<double>(g, oT, i, gather, n), mg_slice<double>(g, oS, i, gather, n));
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
    }
    return mg_gather_outputs(g, gather, n, {oQ, oT, oS});
}

int dxo_mgpu_von_mises(dxo_mgpu* g, const dxo_vm_params* prm, int d, int64_t n_per_rank, int gather,
                       const double* const* deps, const double* const* sigma_n, const double* const* p,
                       double* const* C_tang, double* const* sigma, double* const* dp) {
    if (!g) return DXO_E_NULL;
    if (!prm || !deps || !sigma_n || !p || !C_tang || !sigma || !dp) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_von_mises: NULL argument");
    if (d != 4 && d != 6) return mg_fail(g, DXO_E_DIM, "dxo_mgpu_von_mises: d must be 4 or 6");
    if (n_per_rank < 0) return mg_fail(g, DXO_E_SIZE, "dxo_mgpu_von_mises: n_per_rank < 0");
    if (gather < DXO_GATHER_NONE || gather > DXO_GATHER_COMPACT_PIPELINED) return mg_fail(g, DXO_E_MEM, "dxo_mgpu_von_mises: bad gather mode");
    if (gather != DXO_GATHER_NONE && (n_per_rank % 2)) return mg_fail(g, DXO_E_ALIGN, "dxo_mgpu_von_mises: with a gather n_per_rank must be even (16-byte aligned blocks)");
    const size_t n = (size_t)n_per_rank, L = g->ctx.size();
    const bool compact = gather >= DXO_GATHER_COMPACT;
    if (compact)
        for (size_t i = 0; i < L; ++i)
            if (!C_tang[i] || !sigma[i] || !dp[i]) return mg_fail(g, DXO_E_NULL, "dxo_mgpu_von_mises: NULL output array");
    // the return map of points [b, e) of every local device's block, written into its slice of the full-length outputs. COMPACT
    // forms: (sigma, dp) only, with the 0/0 point marked in the sign bit of dp — every tangent is rebuilt afterwards
    auto kernels = [&](size_t b, size_t e) -> int {
        for (size_t i = 0; i < L; ++i) {
            const size_t off = (gather == DXO_GATHER_NONE ? 0 : (size_t)g->rank[i] * n) + b;
            int rc;
            {   // option switch, call and restore as ONE unit under the context's (recursive) lock: another thread calling
                // dxo_von_mises on the same borrowed context never sees marks it did not ask for
                DXO_LOCK(g->ctx[i]);
                const int64_t saved_mark = g->ctx[i]->vm_mark_indeterminate;
                if (compact) g->ctx[i]->vm_mark_indeterminate = 1;
                rc = dxo_von_mises(g->ctx[i], prm, d, (int64_t)(e - b), DXO_MEM_DEVICE, deps[i] + b * d, sigma_n[i] + b * d, p[i] + b,
                                   compact ? nullptr : (C_tang[i] ? C_tang[i] + off * d * d : nullptr),
                                   sigma[i] ? sigma[i] + off * d : nullptr, dp[i] ? dp[i] + off : nullptr);
                g->ctx[i]->vm_mark_indeterminate = saved_mark;
            }
            if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
        }
        return DXO_OK;
    };
    // the tangents of points [b, e) of EVERY rank's block from the gathered state, on each local device's launch stream
    auto rebuild = [&](size_t b, size_t e) -> int {
        for (size_t i = 0; i < L; ++i)
            for (int r = 0; r < g->world; ++r) {
                const size_t o = (size_t)r * n + b;
                const int rc = dxo_vm_expand_tangent(g->ctx[i], prm, d, (int64_t)(e - b), DXO_MEM_DEVICE, sigma[i] + o * d, dp[i] + o, C_tang[i] + o * d * d);
                if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
            }
        return DXO_OK;
    };
    auto clear_marks = [&]() -> int {
        for (size_t i = 0; i < L; ++i) {
            const int rc = dxo_vm_clear_marks(g->ctx[i], (int64_t)((size_t)g->world * n), dp[i]);
            if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
        }
        return DXO_OK;
    };
    int rc;
    if (gather == DXO_GATHER_COMPACT_PIPELINED && n > 0) {
        // the block in `chunks` pieces on 64-point borders; the kernel of piece k + 1 runs while piece k is on the
        // links (exchange stream), and piece k's tangents are rebuilt while later pieces are still travelling
        int64_t chunks = g->ctx[0]->mgpu_chunks;
        if (chunks < 1) chunks = 1;
        if (chunks > 64) chunks = 64;
        size_t step = (n + (size_t)chunks - 1) / (size_t)chunks;
        step = (step + DXO_WAVE - 1) / DXO_WAVE * DXO_WAVE;
        const int nk = (int)((n + step - 1) / step);
        rc = mg_pipeline_resources(g, nk);
        if (rc != DXO_OK) return rc;
        for (int k = 0; k < nk; ++k) {
            const size_t b = (size_t)k * step, e = b + step < n ? b + step : n;
            rc = kernels(b, e);
            if (rc != DXO_OK) return rc;
            if (g->world == 1) continue;
            for (size_t i = 0; i < L; ++i) {
                (void)hipSetDevice(g->ctx[i]->device);
                hipError_t he = hipEventRecord(g->ev_kernel[i * g->ev_chunks + k], dxo_launch_stream(g->ctx[i]));
                if (he == hipSuccess) he = hipStreamWaitEvent(g->xstream[i], g->ev_kernel[i * g->ev_chunks + k], 0);
                if (he != hipSuccess) return mg_hip_fail(g, he, "dxo_mgpu_von_mises: event record / wait failed");
            }
            rc = mg_exchange_direct(g, (void* const*)sigma, sizeof(double) * (size_t)d, n, b, e - b, &g->xstream);
            if (rc == DXO_OK) rc = mg_exchange_direct(g, (void* const*)dp, sizeof(double), n, b, e - b, &g->xstream);
            if (rc != DXO_OK) return rc;
            for (size_t i = 0; i < L; ++i) {
                (void)hipSetDevice(g->ctx[i]->device);
                const hipError_t he = hipEventRecord(g->ev_arrived[i * g->ev_chunks + k], g->xstream[i]);
                if (he != hipSuccess) return mg_hip_fail(g, he, "dxo_mgpu_von_mises: event record failed");
            }
        }
        for (int k = 0; k < nk; ++k) {
            const size_t b = (size_t)k * step, e = b + step < n ? b + step : n;
            if (g->world > 1)
                for (size_t i = 0; i < L; ++i) {
                    (void)hipSetDevice(g->ctx[i]->device);
                    const hipError_t he = hipStreamWaitEvent(dxo_launch_stream(g->ctx[i]), g->ev_arrived[i * g->ev_chunks + k], 0);
                    if (he != hipSuccess) return mg_hip_fail(g, he, "dxo_mgpu_von_mises: event wait failed");
                }
            rc = rebuild(b, e);
            if (rc != DXO_OK) return rc;
        }
        return clear_marks();
    }
    // 1. every local device: return map of its own cell block
    rc = kernels(0, n);
    if (rc != DXO_OK) return rc;
    // 2. the exchange step (a world of one has none; the COMPACT forms still owe the tangent of the only block)
    if (gather == DXO_GATHER_NONE || n == 0) return DXO_OK;
    if (g->world > 1) {
        if (gather == DXO_GATHER_COMPACT_DIRECT) {
            rc = mg_exchange_direct(g, (void* const*)sigma, sizeof(double) * (size_t)d, n, 0, n, nullptr);
            if (rc == DXO_OK) rc = mg_exchange_direct(g, (void* const*)dp, sizeof(double), n, 0, n, nullptr);
            if (rc != DXO_OK) return rc;
        } else {
            rc = dxo_mgpu_all_gather(g, sigma, n_per_rank * d);
            if (rc == DXO_OK) rc = dxo_mgpu_all_gather(g, dp, n_per_rank);
            if (rc != DXO_OK) return rc;
            if (gather == DXO_GATHER_FULL) return dxo_mgpu_all_gather(g, C_tang, n_per_rank * d * d);
        }
    }
    if (!compact) return DXO_OK;
    // 3. COMPACT forms: the tangent of EVERY block (own block included) from the gathered state, on each device's stream
    //    behind its exchange: one launch over the full range, then the marks are cleared
    for (size_t i = 0; i < L; ++i) {
        const int64_t all = (int64_t)((size_t)g->world * n);
        rc = dxo_vm_expand_tangent(g->ctx[i], prm, d, all, DXO_MEM_DEVICE, sigma[i], dp[i], C_tang[i]);
        if (rc != DXO_OK) return mg_fail(g, rc, dxo_last_error(g->ctx[i]));
    }
    return clear_marks();
}

}  // extern "C"
