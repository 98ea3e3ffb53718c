// heat_field.hip — operand evaluation fused in front of the nonlinear heat flux (BASELINE config 1).
//
// The reference evaluates the operands T and sigma = grad T with Expression.eval and hands the two arrays to
// q_impl / dqdT_impl / dqdsigma_impl (doc/demo/demo_nonlinear_heat_equation_part2.py:219-261, :299-309). Here a wave
// gathers its cells' temperature dofs, each lane forms (T, grad T) of its own quadrature point (operand_core.h,
// kind VALUE_GRAD) and applies k = 1/(A + B T), q = -k sigma, dq/dT = B k^2 sigma, dq/dsigma = -k I; the results leave
// through the wave's LDS slice in output order. The operand arrays never exist in memory.
#include "dxo_common.h"
#include "operand_core.h"

namespace {

template <int G>
__global__ __launch_bounds__(DXO_BLOCK) void heat_field(double A, double B, OperandDev m, int64_t cell0, int64_t n_cells,
                                                        const double* __restrict__ Td, double* __restrict__ q,
                                                        double* __restrict__ dqdT, double* __restrict__ dqds) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * m.wave_doubles;
    const int cpw = m.cells_per_wave;
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = (n_cells - c0 < cpw) ? (int)(n_cells - c0) : cpw;
        const int npts = ncell * m.nq;
        const int64_t p0 = c0 * m.nq;
        double o[1 + G];
        const bool active = operand_point<G, 1, DXO_OPERAND_VALUE_GRAD>(m, tab, W, Td, nullptr, cell0 + c0, ncell, lane, o);
        const double k = 1.0 / (A + B * o[0]);       // :216
        const double mk = -k, bk2 = B * (k * k);     // :228, :246
        // three passes through the wave's LDS slice, each stored in output order
        if (q) {
            if (active)
#pragma unroll
                for (int a = 0; a < G; ++a) W[lane * G + a] = mk * o[1 + a];
            op_fence();
            for (int idx = lane; idx < npts * G; idx += DXO_WAVE) __builtin_nontemporal_store(W[idx], q + p0 * G + idx);
            op_fence();
        }
        if (dqdT) {
            if (active)
#pragma unroll
                for (int a = 0; a < G; ++a) W[lane * G + a] = bk2 * o[1 + a];
            op_fence();
            for (int idx = lane; idx < npts * G; idx += DXO_WAVE) __builtin_nontemporal_store(W[idx], dqdT + p0 * G + idx);
            op_fence();
        }
        if (dqds) {
            if (active)
#pragma unroll
                for (int a = 0; a < G; ++a)
#pragma unroll
                    for (int b = 0; b < G; ++b) W[(lane * G + a) * G + b] = a == b ? mk : 0.0;   // :260
            op_fence();
            for (int idx = lane; idx < npts * G * G; idx += DXO_WAVE) __builtin_nontemporal_store(W[idx], dqds + p0 * G * G + idx);
            op_fence();
        }
    }
}

struct HeatFieldLaunch {
    double A, B;
    const dxo_mesh* mesh;
    const double* d_T;
    bool has_q, has_dT, has_ds;
    int64_t next_cell;
};

int heat_field_launch(dxo_ctx* ctx, const HeatFieldLaunch& L, int64_t cell0, int64_t n_cells, double* q, double* dqdT,
                      double* dqds, hipStream_t s) {
    if (n_cells == 0) return DXO_OK;
    const OperandDev& m = L.mesh->dev;
    const size_t shm = (size_t)(m.table_doubles + 4 * m.wave_doubles) * sizeof(double);
    const int64_t n_groups = (n_cells + m.cells_per_wave - 1) / m.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;
    if (L.mesh->gdim == 2)
        hipLaunchKernelGGL(heat_field<2>, dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.A, L.B, m, cell0, n_cells, L.d_T, q, dqdT, dqds);
    else
        hipLaunchKernelGGL(heat_field<3>, dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.A, L.B, m, cell0, n_cells, L.d_T, q, dqdT, dqds);
    return DXO_OK;
}

int heat_field_chunk(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const*, void* const* d_out, hipStream_t s) {
    HeatFieldLaunch& L = *static_cast<HeatFieldLaunch*>(user);
    const int64_t cell0 = L.next_cell;
    L.next_cell += n_chunk;
    int o = 0;
    double* q = L.has_q ? (double*)d_out[o++] : nullptr;
    double* dT = L.has_dT ? (double*)d_out[o++] : nullptr;
    double* ds = L.has_ds ? (double*)d_out[o++] : nullptr;
    return heat_field_launch(ctx, L, cell0, n_chunk, q, dT, ds, s);
}

}  // namespace

extern "C" int dxo_heat_field(dxo_ctx* ctx, double A, double B, dxo_mesh* mesh, int mem, const double* T_dofs, double* q,
                              double* dqdT, double* dqdsigma) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_heat_field: mesh is NULL");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_heat_field: bad mem");
    const int64_t nc = mesh->num_cells;
    if (nc == 0 || (!q && !dqdT && !dqdsigma)) return DXO_OK;
    if (!T_dofs) return dxo_fail(ctx, DXO_E_NULL, "dxo_heat_field: T_dofs is NULL");
    if (((uintptr_t)T_dofs | (uintptr_t)q | (uintptr_t)dqdT | (uintptr_t)dqdsigma) & 7u)
        return dxo_fail(ctx, DXO_E_ALIGN, "dxo_heat_field: arrays must be 8-byte aligned");
    const int G = mesh->gdim, nq = mesh->dev.nq;
    HeatFieldLaunch L{A, B, mesh, T_dofs, q != nullptr, dqdT != nullptr, dqdsigma != nullptr, 0};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = heat_field_launch(ctx, L, 0, nc, q, dqdT, dqdsigma, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    const size_t tb = (size_t)mesh->num_field_nodes * sizeof(double);
    if (mesh->u_cap < tb) {
        if (mesh->d_u) DXO_HIP(ctx, hipFree(mesh->d_u));
        mesh->d_u = nullptr;
        mesh->u_cap = 0;
        DXO_HIP(ctx, hipMalloc((void**)&mesh->d_u, tb));
        mesh->u_cap = tb;
    }
    DXO_HIP(ctx, hipMemcpy(mesh->d_u, T_dofs, tb, hipMemcpyHostToDevice));
    L.d_T = mesh->d_u;
    const size_t sd = sizeof(double) * (size_t)nq;
    std::vector<dxo_span> in;
    std::vector<dxo_span> out;
    if (q) out.push_back({nullptr, q, G * sd});
    if (dqdT) out.push_back({nullptr, dqdT, G * sd});
    if (dqdsigma) out.push_back({nullptr, dqdsigma, G * G * sd});
    return dxo_run_host_pipeline(ctx, nc, in, out, heat_field_chunk, &L, nq, nullptr, true);
}
