// Does vector work overlap with fp32-input MFMA on gfx950? One wave (or two) per SIMD runs a loop of independent
// v_mfma_f32_32x32x2_f32 with N plain fp32 VALU instructions (v_fma_f32) or N transcendental ones (v_exp_f32) per MFMA.
// If the matrix pipe is separate, time per MFMA stays 64 cycles until the N instructions no longer fit the gap; if the
// fp32 MFMA shares the vector FMA lanes, it grows from N = 1.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma32_valu_probe mfma32_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

// the same loop on the bf16 pipe: v_mfma_f32_32x32x16_bf16 (32 cycles per SIMD), N vector instructions after each
template <int N, int KIND>
__global__ __launch_bounds__(512, 2) void probe_bf16(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + k + threadIdx.x * 1e-3f;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(seed * 0.5f); b[k] = (__bf16)(seed * 0.25f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
                else v[j & 7] = __builtin_amdgcn_exp2f(v[j & 7] * 0.001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 16; ++q) s += acc[k][q];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int KIND>
__global__ __launch_bounds__(512, 2) void probe(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + k + threadIdx.x * 1e-3f;
    const float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
                else v[j & 7] = __builtin_amdgcn_exp2f(v[j & 7] * 0.001f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k)
        for (int q = 0; q < 16; ++q) s += acc[k][q];
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int KIND>
double run(int waves_per_simd, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe<N, KIND>), dim3(256), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<N, KIND>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // ns per MFMA per SIMD: each SIMD runs waves_per_simd waves, each 4*iters MFMAs
    return ms * 1e6 / (4.0 * iters * waves_per_simd);
}

template <int N, int KIND>
double run_bf16(int waves_per_simd, float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((probe_bf16<N, KIND>), dim3(256), dim3(threads), 0, 0, d, 10, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe_bf16<N, KIND>), dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (4.0 * iters * waves_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    printf("ns per MFMA per SIMD (64 cycles at 2.4 GHz = 26.7 ns). columns: N extra instr per MFMA; rows: kind x waves/SIMD\n");
#define ROW(KIND, W) printf("%s w=%d:", KIND ? "v_exp_f32" : "v_fma_f32", W); \
    printf(" N0 %.1f", run<0, KIND>(W, d, iters)); printf(" N1 %.1f", run<1, KIND>(W, d, iters)); printf(" N2 %.1f", run<2, KIND>(W, d, iters)); \
    printf(" N4 %.1f", run<4, KIND>(W, d, iters)); printf(" N8 %.1f", run<8, KIND>(W, d, iters)); printf(" N12 %.1f", run<12, KIND>(W, d, iters)); \
    printf(" N16 %.1f", run<16, KIND>(W, d, iters)); printf(" N24 %.1f\n", run<24, KIND>(W, d, iters));
    ROW(0, 1) ROW(0, 2) ROW(1, 1) ROW(1, 2)
    printf("bf16 pipe, v_mfma_f32_32x32x16_bf16 (32 cycles at 2.4 GHz = 13.3 ns):\n");
#define ROWB(KIND, W) printf("%s w=%d:", KIND ? "v_exp_f32" : "v_fma_f32", W); \
    printf(" N0 %.1f", run_bf16<0, KIND>(W, d, iters)); printf(" N1 %.1f", run_bf16<1, KIND>(W, d, iters)); printf(" N2 %.1f", run_bf16<2, KIND>(W, d, iters)); \
    printf(" N4 %.1f", run_bf16<4, KIND>(W, d, iters)); printf(" N8 %.1f", run_bf16<8, KIND>(W, d, iters)); printf(" N12 %.1f", run_bf16<12, KIND>(W, d, iters)); \
    printf(" N16 %.1f", run_bf16<16, KIND>(W, d, iters)); printf(" N24 %.1f\n", run_bf16<24, KIND>(W, d, iters));
    ROWB(0, 1) ROWB(0, 2) ROWB(1, 1) ROWB(1, 2)
    return 0;
}
