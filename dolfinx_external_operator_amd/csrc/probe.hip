// probe.hip — HBM stream probe: the roof the pointwise kernels are measured against.
//
// A kernel with NO arithmetic that moves the same read : write mix as a constitutive kernel, all
// accesses lane-linear 16 bytes: per tile each lane loads R and stores W 16-byte chunks
// (R = 13, W = 43 is the von Mises d = 6 mix of 13 doubles in / 43 doubles out per point; R = W = 1 is
// a plain copy). Its GB/s is the practically reachable ceiling for that mix on this chip, reported next
// to the 8 TB/s spec peak ("also report against a measured device-copy bandwidth").
#include "dxo_common.h"

namespace {

template <int R, int W, bool NT>
__global__ __launch_bounds__(DXO_BLOCK) void stream_probe(int64_t n_tiles, const dxo_f64x2* __restrict__ src,
                                                          dxo_f64x2* __restrict__ dst) {
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int WAVES = DXO_BLOCK / DXO_WAVE;
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    for (int64_t t = (int64_t)blockIdx.x * WAVES + wave; t < n_tiles; t += stride) {
        const dxo_f64x2* s = src + t * (R * DXO_WAVE);
        dxo_f64x2* d = dst + t * (W * DXO_WAVE);
        dxo_f64x2 acc = {0.0, 0.0};
        dxo_f64x2 v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = s[k * DXO_WAVE + lane];
#pragma unroll
        for (int k = 0; k < R; ++k) acc += v[k];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const dxo_f64x2 o = acc + v[k % R];
            if constexpr (NT)
                __builtin_nontemporal_store(o, d + k * DXO_WAVE + lane);
            else
                d[k * DXO_WAVE + lane] = o;
        }
    }
}

template <int R, int W>
void launch(dxo_ctx* ctx, int64_t n_tiles, const void* src, void* dst, hipStream_t s) {
    const int grid = dxo_grid_for_tiles(ctx, n_tiles, DXO_BLOCK / DXO_WAVE);
    if (ctx->nontemporal)
        hipLaunchKernelGGL((stream_probe<R, W, true>), dim3(grid), dim3(DXO_BLOCK), 0, s, n_tiles, (const dxo_f64x2*)src, (dxo_f64x2*)dst);
    else
        hipLaunchKernelGGL((stream_probe<R, W, false>), dim3(grid), dim3(DXO_BLOCK), 0, s, n_tiles, (const dxo_f64x2*)src, (dxo_f64x2*)dst);
}

}  // namespace

extern "C" int dxo_stream_probe(dxo_ctx* ctx, int read_chunks, int write_chunks, int64_t n_tiles, const void* src,
                                void* dst) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (n_tiles < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_stream_probe: n_tiles < 0");
    if (n_tiles > 0 && (!src || !dst)) return dxo_fail(ctx, DXO_E_NULL, "dxo_stream_probe: NULL buffer");
    if (((uintptr_t)src | (uintptr_t)dst) & 15u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_stream_probe: 16-byte alignment");
    hipStream_t s = dxo_launch_stream(ctx);
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    if (n_tiles > 0) {
        if (read_chunks == 13 && write_chunks == 43) launch<13, 43>(ctx, n_tiles, src, dst, s);
        else if (read_chunks == 9 && write_chunks == 21) launch<9, 21>(ctx, n_tiles, src, dst, s);
        else if (read_chunks == 8 && write_chunks == 20) launch<8, 20>(ctx, n_tiles, src, dst, s);
        else if (read_chunks == 3 && write_chunks == 8) launch<3, 8>(ctx, n_tiles, src, dst, s);
        else if (read_chunks == 1 && write_chunks == 1) launch<1, 1>(ctx, n_tiles, src, dst, s);
        else if (read_chunks == 4 && write_chunks == 4) launch<4, 4>(ctx, n_tiles, src, dst, s);
        else return dxo_fail(ctx, DXO_E_DIM, "dxo_stream_probe: unsupported (read, write) mix");
    }
    return dxo_device_end(ctx, s);
}
