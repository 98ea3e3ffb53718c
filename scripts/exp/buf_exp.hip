// buf_exp.hip — does streaming-write bandwidth depend on WHICH allocation / address offset is written?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void pure_write(long n_chunks, f64x2* __restrict__ dst) {
    const long stride = (long)gridDim.x * 256 * 8;
    for (long i = ((long)blockIdx.x * 256 * 8) + threadIdx.x; i < n_chunks; i += stride) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const f64x2 o = {(double)i, (double)k}; if (i + k * 256 < n_chunks) { if (NT) __builtin_nontemporal_store(o, dst + i + k * 256); else dst[i + k * 256] = o; } }
    }
}
__global__ __launch_bounds__(256) void pure_read(long n_chunks, const f64x2* __restrict__ src, f64x2* __restrict__ sink) {
    const long stride = (long)gridDim.x * 256 * 8;
    f64x2 acc = {0, 0};
    for (long i = ((long)blockIdx.x * 256 * 8) + threadIdx.x; i < n_chunks; i += stride) {
        f64x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (i + k * 256 < n_chunks) ? src[i + k * 256] : f64x2{0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    if (acc.x == 1.2345e301) sink[threadIdx.x] = acc;
}
float timeit(hipStream_t st, int launches, const std::function<void()>& fn);
#include <functional>
float timeit(hipStream_t st, int launches, const std::function<void()>& fn) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> v;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, st));
        for (int l = 0; l < launches; ++l) fn();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); v.push_back(ms / launches);
    }
    std::sort(v.begin(), v.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return v[2];
}
int main(int argc, char** argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 8;
    const size_t bytes = (argc > 2 ? atol(argv[2]) : 3000L) * 1000000L;
    hipStream_t st; CK(hipStreamCreate(&st));
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("free %.1f GB total %.1f GB\n", fr / 1e9, tot / 1e9);
    std::vector<char*> bufs(nbuf);
    f64x2* sink; CK(hipMalloc(&sink, 4096));
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&bufs[i], bytes + (64 << 20))); CK(hipMemset(bufs[i], 0, bytes)); }
    const long chunks = bytes / 16;
    const int grid = (int)((chunks + 2047) / 2048);
    for (int i = 0; i < nbuf; ++i) {
        f64x2* p = (f64x2*)bufs[i];
        float w = timeit(st, 5, [&] { hipLaunchKernelGGL((pure_write<true>), dim3(grid), dim3(256), 0, st, chunks, p); });
        float w2 = timeit(st, 5, [&] { hipLaunchKernelGGL((pure_write<false>), dim3(grid), dim3(256), 0, st, chunks, p); });
        float r = timeit(st, 5, [&] { hipLaunchKernelGGL(pure_read, dim3(grid), dim3(256), 0, st, chunks, (const f64x2*)p, sink); });
        printf("buf %d va %p  writeNT %7.1f GB/s  write %7.1f GB/s  read %7.1f GB/s\n", i, (void*)p, bytes / w / 1e6, bytes / w2 / 1e6, bytes / r / 1e6);
    }
    // offsets inside buffer 0
    const long offs[] = {0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 33 << 20};
    for (long off : offs) {
        f64x2* p = (f64x2*)(bufs[0] + off);
        float w = timeit(st, 5, [&] { hipLaunchKernelGGL((pure_write<true>), dim3(grid), dim3(256), 0, st, chunks, p); });
        printf("buf 0 + %9ld  writeNT %7.1f GB/s\n", off, bytes / w / 1e6);
    }
    // sizes
    for (size_t mb : {256, 512, 1000, 2000, 3000}) {
        const long ch = mb * 1000000L / 16;
        for (int i = 0; i < 2; ++i) {
            f64x2* p = (f64x2*)bufs[i];
            float w = timeit(st, 5, [&] { hipLaunchKernelGGL((pure_write<true>), dim3((int)((ch + 2047) / 2048)), dim3(256), 0, st, ch, p); });
            printf("buf %d size %5zu MB writeNT %7.1f GB/s\n", i, mb, mb * 1e6 / w / 1e6);
        }
    }
    return 0;
}
