// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand/result lane layout and issue rate.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/exp/mfma64_probe scripts/exp/mfma64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void layout(const double* A, const double* B, double* D) {   // A[16][4], B[4][16] row-major, D[16][16]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];
    const double b = B[(l / 16) * 16 + l % 16];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];   // raw dump: lane-major
}

__global__ void rate(double* out, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-9, b = 1.0 - l * 1e-9;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    std::vector<double> A(64), B(64), D(256);
    // D[i][j] = sum_k A[i][k] B[k][j] = (i+1) + 1000 (j+1) + 1e6 * [k=2 term] + 1e9 * [k=3 term]: unique per (i, j),
    // and every k slot contributes, so a wrong A/B lane layout cannot go unnoticed
    for (int i = 0; i < 16; ++i) { A[i * 4 + 0] = i + 1; A[i * 4 + 1] = 1; A[i * 4 + 2] = 1e6; A[i * 4 + 3] = 0.5; }
    for (int j = 0; j < 16; ++j) { B[0 * 16 + j] = 1; B[1 * 16 + j] = 1000.0 * (j + 1); B[2 * 16 + j] = 1; B[3 * 16 + j] = 2e9; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = D[l * 4 + r];
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(v - ((i + 1) + 1000.0 * (j + 1) + 1e6 + 1e9)) < 1e-3) { fi = i; fj = j; }
        const int ei = 4 * (l / 16) + r, ej = l % 16;
        if (fi != ei || fj != ej) ok = 0;
        if (l % 16 < 2 || !ok) { if (l < 34) printf("lane %2d reg %d holds D[%2d][%2d]  (raw %.1f)\n", l, r, fi, fj, v); }
    }
    printf("layout guess D[4*(l/16)+r][l%%16], A[l%%16][l/16], B[l/16][l%%16]: %s\n", ok ? "CONFIRMED" : "WRONG");
    double* dout; hipMalloc(&dout, 256 * 1024 * 256 * 8);
    const int iters = 2000;
    for (int waves : {4, 8, 16}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rate, dim3(256 * 4), dim3(64 * waves / 4), 0, 0, dout, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate, dim3(256 * 4), dim3(64 * waves / 4), 0, 0, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = 256.0 * 4 * (waves / 4.0) * iters * 4;     // wave-level MFMA instructions
        printf("blocks 1024 x %d waves: %.3f ms, %.1f TFLOP/s fp64, %.1f clk/MFMA/SIMD at 2.4 GHz\n", waves / 4, ms,
               n_mfma * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n_mfma / 1024));
    }
    return 0;
}
