/* c_abi_demo.c — libdxo_hip.so used from plain C: no Python, no torch, no HIP headers.
 *
 * What a compiled host program (or a cgo/JNI/FFI binding) does to replace the reference's per-point von Mises
 * kernel (doc/demo/demo_plasticity_von_mises.py:298-352): create a context, hand over host arrays, read the results
 * back in the reference's order (C_tang, sigma, dp); then the same with the state resident on the device and the
 * history update of :564-565 done there.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o examples/c_abi_demo \
 *       -Ldolfinx_external_operator_amd -ldxo_hip -Wl,-rpath,'$ORIGIN/../dolfinx_external_operator_amd' -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dxo.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? dxo_last_error(ctx) : ""); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(void) {
    dxo_ctx* ctx = NULL;
    int n_dev = 0;
    if (dxo_device_count(&n_dev) != 0 || n_dev < 1) {
        fprintf(stderr, "no HIP device: libdxo_hip has no CPU path\n");
        return 2;
    }
    CHECK(dxo_ctx_create(0, &ctx));
    const int d = 4;
    const int64_t n = 100000;
    const double E = 70e3, nu = 0.3, Et = E / 100.0;
    dxo_vm_params prm = {E, nu, 250.0, E * Et / (E - Et)};                 /* :185-188 */
    double* deps = malloc(n * d * sizeof(double));
    double* sigma_n = calloc(n * d, sizeof(double));
    double* p = calloc(n, sizeof(double));
    double* C_tang = malloc(n * d * d * sizeof(double));
    double* sigma = malloc(n * d * sizeof(double));
    double* dp = malloc(n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) {                                      /* uniaxial strain ramp */
        deps[i * d + 0] = 1e-5 * (double)(i % 1000);
        deps[i * d + 1] = deps[i * d + 2] = deps[i * d + 3] = 0.0;
    }
    CHECK(dxo_von_mises(ctx, &prm, d, n, DXO_MEM_HOST, deps, sigma_n, p, C_tang, sigma, dp));
    int64_t plastic = 0;
    double worst = 0.0;
    const double mu = E / (2 * (1 + nu));
    for (int64_t i = 0; i < n; ++i) {
        if (dp[i] <= 0.0) continue;
        ++plastic;
        const double* s = sigma + i * d;
        const double m = (s[0] + s[1] + s[2]) / 3.0;
        const double seq = sqrt(1.5 * ((s[0] - m) * (s[0] - m) + (s[1] - m) * (s[1] - m) + (s[2] - m) * (s[2] - m) + s[3] * s[3]));
        const double f = fabs(seq - prm.sigma_0 - prm.H * dp[i]);          /* yield condition after the return */
        if (f > worst) worst = f;
    }
    printf("host arrays : %lld points, %lld plastic, max |f(sigma, p+dp)| = %.3e, C_tang[0][0][0] = %.6f (lambda + 2 mu = %.6f)\n",
           (long long)n, (long long)plastic, worst, C_tang[0], E * nu / ((1 + nu) * (1 - 2 * nu)) + 2 * mu);
    if (worst > 1e-8 * prm.sigma_0 || plastic == 0) return 3;

    /* the same batch with the state on the device and the history update done there */
    void *d_deps, *d_sn, *d_p, *d_C, *d_s, *d_dp;
    CHECK(dxo_device_alloc(ctx, n * d * 8, &d_deps));
    CHECK(dxo_device_alloc(ctx, n * d * 8, &d_sn));
    CHECK(dxo_device_alloc(ctx, n * 8, &d_p));
    CHECK(dxo_device_alloc(ctx, n * d * d * 8, &d_C));
    CHECK(dxo_device_alloc(ctx, n * d * 8, &d_s));
    CHECK(dxo_device_alloc(ctx, n * 8, &d_dp));
    CHECK(dxo_copy(ctx, d_deps, deps, n * d * 8, 0));
    CHECK(dxo_copy(ctx, d_sn, sigma_n, n * d * 8, 0));
    CHECK(dxo_copy(ctx, d_p, p, n * 8, 0));
    CHECK(dxo_von_mises(ctx, &prm, d, n, DXO_MEM_DEVICE, d_deps, d_sn, d_p, d_C, d_s, d_dp));
    CHECK(dxo_vm_commit_state(ctx, d, n, d_p, d_dp, d_sn, d_s));          /* p += dp; sigma_n <- sigma */
    CHECK(dxo_ctx_synchronize(ctx));
    double* p_back = malloc(n * sizeof(double));
    double* sn_back = malloc(n * d * sizeof(double));
    CHECK(dxo_copy(ctx, p_back, d_p, n * 8, 1));
    CHECK(dxo_copy(ctx, sn_back, d_sn, n * d * 8, 1));
    int same = memcmp(p_back, dp, n * 8) == 0 && memcmp(sn_back, sigma, n * d * 8) == 0;   /* p was 0, so p == dp */
    printf("device state: history update %s the host result\n", same ? "reproduces" : "DIFFERS FROM");
    /* host arrays again, but the history variables live in a device mirror (dxo_vm_state): uploaded once, every call
     * sends deps only, and the load-step update is applied on the device while the caller applies it to its arrays */
    dxo_vm_state* st = NULL;
    CHECK(dxo_vm_state_create(ctx, d, n, &st));
    CHECK(dxo_vm_state_upload(ctx, st, DXO_MEM_HOST, sigma_n, p));
    double* C2 = malloc(n * d * d * sizeof(double));
    double* s2 = malloc(n * d * sizeof(double));
    double* dp2 = malloc(n * sizeof(double));
    CHECK(dxo_von_mises_state(ctx, &prm, st, DXO_MEM_HOST, deps, C2, s2, dp2));
    int same_state = memcmp(C2, C_tang, n * d * d * 8) == 0 && memcmp(s2, sigma, n * d * 8) == 0 && memcmp(dp2, dp, n * 8) == 0;
    CHECK(dxo_vm_state_commit(ctx, st));                                   /* p += dp; sigma_n <- sigma, in the mirror */
    CHECK(dxo_vm_state_download(ctx, st, DXO_MEM_HOST, sn_back, p_back));
    same_state = same_state && memcmp(p_back, dp, n * 8) == 0 && memcmp(sn_back, sigma, n * d * 8) == 0;
    printf("resident state: call and commit %s the plain call\n", same_state ? "reproduce" : "DIFFER FROM");
    dxo_vm_state_destroy(ctx, st);
    free(C2); free(s2); free(dp2);
    same = same && same_state;
    dxo_device_free(ctx, d_deps); dxo_device_free(ctx, d_sn); dxo_device_free(ctx, d_p);
    dxo_device_free(ctx, d_C); dxo_device_free(ctx, d_s); dxo_device_free(ctx, d_dp);
    free(deps); free(sigma_n); free(p); free(C_tang); free(sigma); free(dp); free(p_back); free(sn_back);
    dxo_ctx_destroy(ctx);
    return same ? 0 : 4;
}
