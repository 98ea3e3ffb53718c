// mock_rccl.cpp — a stand-in for librccl.so.1, TEST ONLY (tests/test_mgpu_mock_transport.py).
//
// The builder's GPU box has one MI355X and RCCL refuses two ranks on one device, so the N > 1 data path of libdxo
// (csrc/mgpu.hip: in-place all-gather, the direct send / receive exchange, the chunked overlap form) has never moved a byte
// between ranks. This library implements the handful of RCCL entry points libdxo resolves with dlsym — for ONE process whose
// "devices" may all be the same GPU (dxo_mgpu_create with DXO_MGPU_TEST_SHARE_DEVICE=1): operations are collected between
// ncclGroupStart / ncclGroupEnd and carried out at ncclGroupEnd as device-to-device copies on the receiving rank's stream, after
// every stream of the group has been synchronised (a transport without any overlap — the test is about WHICH bytes land WHERE).
// It is not RCCL, measures nothing and is never loaded by the product: libdxo finds it only when the test puts its directory in
// front of LD_LIBRARY_PATH in a process that has not loaded the real library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;

struct MockComm {
    int rank, world, device;
};
typedef MockComm* ncclComm_t;

}  // extern "C"

namespace {

struct Op {
    int kind;            // 0 all-gather, 1 send, 2 recv
    MockComm* comm;
    const void* src;
    void* dst;
    size_t bytes;
    int peer;
    hipStream_t stream;
    bool done = false;
};

std::vector<Op> g_ops;
int g_depth = 0;
long g_bytes_moved = 0, g_groups = 0;

ncclResult_t run_group() {
    ++g_groups;
    for (const Op& o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;      // producers have finished
    // all-gathers: every rank's send buffer into every rank's receive buffer at rank * bytes
    for (const Op& d : g_ops) {
        if (d.kind != 0) continue;
        for (const Op& s : g_ops) {
            if (s.kind != 0 || s.bytes != d.bytes) continue;
            char* to = static_cast<char*>(d.dst) + (size_t)s.comm->rank * d.bytes;
            if (to != s.src) {
                if (hipMemcpyAsync(to, s.src, d.bytes, hipMemcpyDeviceToDevice, d.stream) != hipSuccess) return ncclUnhandledCudaError;
                g_bytes_moved += (long)d.bytes;
            }
        }
    }
    // point to point: the k-th send of rank a to rank b meets the k-th receive of rank b from rank a
    for (Op& r : g_ops) {
        if (r.kind != 2) continue;
        bool matched = false;
        for (Op& s : g_ops) {
            if (s.kind != 1 || s.done || s.peer != r.comm->rank || s.comm->rank != r.peer) continue;
            if (s.bytes != r.bytes) { std::fprintf(stderr, "mock_rccl: send of %zu bytes meets a receive of %zu\n", s.bytes, r.bytes); return ncclInvalidArgument; }
            if (hipMemcpyAsync(r.dst, s.src, r.bytes, hipMemcpyDeviceToDevice, r.stream) != hipSuccess) return ncclUnhandledCudaError;
            g_bytes_moved += (long)r.bytes;
            s.done = r.done = matched = true;
            break;
        }
        if (!matched) { std::fprintf(stderr, "mock_rccl: rank %d receives from %d, nobody sends\n", r.comm->rank, r.peer); return ncclInvalidUsage; }
    }
    for (const Op& s : g_ops)
        if (s.kind == 1 && !s.done) { std::fprintf(stderr, "mock_rccl: rank %d sends to %d, nobody receives\n", s.comm->rank, s.peer); return ncclInvalidUsage; }
    for (const Op& o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    g_ops.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 7, sizeof *id); return ncclSuccess; }

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
    for (int i = 0; i < n; ++i) comms[i] = new MockComm{i, n, devs ? devs[i] : i};
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId, int rank) {
    if (world != 1) return ncclInvalidUsage;       // one process: several ranks only through ncclCommInitAll
    *comm = new MockComm{rank, world, 0};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (--g_depth > 0) return ncclSuccess;
    g_depth = 0;
    return run_group();
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t s) {
    g_ops.push_back(Op{0, c, send, recv, count, -1, s});
    return g_depth ? ncclSuccess : run_group();
}

ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{1, c, send, nullptr, count, peer, s});
    return g_depth ? ncclSuccess : ncclInvalidUsage;
}

ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t s) {
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    g_ops.push_back(Op{2, c, nullptr, recv, count, peer, s});
    return g_depth ? ncclSuccess : ncclInvalidUsage;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success (mock)" : "error (mock_rccl)"; }

// test-side counters
long mock_rccl_bytes_moved() { return g_bytes_moved; }
long mock_rccl_groups() { return g_groups; }

}  // extern "C"
