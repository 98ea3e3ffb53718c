// field_ops.hip — `*_field` entry points of the Mohr-Coulomb, ICNN and analytic Isihara operators: the reference's pair
//   evaluated_operands = evaluate_operands(ops);  evaluate_external_operators(ops, evaluated_operands)
// (demo_plasticity_mohr_coulomb.py:679-688, demo_hyperelasticity.py:548-557) behind ONE call that takes the dof vector
// of the displacement field instead of the operand array: eps(Du) (Mandel, :163-165 of the Mohr-Coulomb demo) or
// F = I + grad u (demo_hyperelasticity.py:479) is formed on the device by the operand kernel (operand.hip) into a staging
// buffer owned by the context and consumed from there by the constitutive kernels.
//
// What this buys: a HOST caller uploads one value per dof instead of one operand tensor per quadrature point (P2
// triangles, 3 points per cell: 16 B per point instead of 32 for the strain; nothing else of the input side crosses
// PCIe for the network operators), and there is no `Expression.eval` on the CPU. Unlike dxo_von_mises_field (vm_field.hip)
// the operand is not formed in the registers of the consuming kernel: the Newton and the network kernels are
// compute-bound (fp64 vector pipe / fp32 MFMA), the 64 B per point that pass through HBM between the two
// launches are a few per cent of their time. The analytic Isihara kernel IS HBM-bound (192 B per point): there F is formed
// in the registers of the kernel that evaluates the model (isihara_field below) and never reaches memory.
//
// Host arrays: the field vector goes up whole, cells stream through the chunked pipeline (units of CELLS), every chunk
// = operand launch + constitutive launch on the chunk's stream.
#include "dxo_common.h"
#include "hyper_core.h"
#include "operand_core.h"

namespace {

// ---- analytic Isihara model with the operand in registers: a wave owns floor(64 / nq) consecutive cells, each lane forms
// F = I + grad u of its own quadrature point (operand_core.h), evaluates P and dP/dF (hyper_core.h) into the wave's LDS
// slice — the gather buffer, free again after the contraction — and the wave stores its points in output order.
template <bool NT>
__global__ __launch_bounds__(DXO_BLOCK) void isihara_field(IsiPrm prm, OperandDev m, int wave_doubles, int64_t cell0, int64_t n_cells,
                                                           const double* __restrict__ u, double* __restrict__ dP,
                                                           double* __restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<2>(m, tab);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * wave_doubles;
    double* Xd = W;                        // [64][16] tangent rows
    double* Xp = W + DXO_WAVE * 16;        // [64][4]  stresses
    const dxo_f64x2* Xd2 = reinterpret_cast<const dxo_f64x2*>(Xd);
    const dxo_f64x2* Xp2 = reinterpret_cast<const dxo_f64x2*>(Xp);
    const int cpw = m.cells_per_wave;
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, DXO_BLOCK / DXO_WAVE, wave);
    for (int64_t grp = walk.first; grp < walk.end; grp += walk.stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = (n_cells - c0 < cpw) ? (int)(n_cells - c0) : cpw;
        const int npts = ncell * m.nq;
        const int64_t p0 = c0 * m.nq;
        double Fv[4];
        const bool active = operand_point<2, 2, DXO_OPERAND_DEFGRAD>(m, tab, W, u, nullptr, cell0 + c0, ncell, lane, Fv);
        if (active) isihara_point(prm, Fv, Xd + lane * 16, Xp + lane * 4);
        op_fence();
        dxo_f64x2* g_d = reinterpret_cast<dxo_f64x2*>(dP + p0 * 16);
        dxo_f64x2* g_p = reinterpret_cast<dxo_f64x2*>(P + p0 * 4);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < npts * 8) {
                if constexpr (NT) __builtin_nontemporal_store(Xd2[idx], g_d + idx);
                else g_d[idx] = Xd2[idx];
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < npts * 2) {
                if constexpr (NT) __builtin_nontemporal_store(Xp2[idx], g_p + idx);
                else g_p[idx] = Xp2[idx];
            }
        }
        op_fence();
    }
}

struct IsiFieldLaunch {
    IsiPrm prm;
    dxo_mesh* mesh;
    const double* d_u;
    int64_t next_cell = 0;
};

int isi_field_launch(dxo_ctx* ctx, const IsiFieldLaunch& L, int64_t cell0, int64_t n_cells, double* dP, double* P, hipStream_t s) {
    if (n_cells == 0) return DXO_OK;
    const OperandDev& m = L.mesh->dev;
    int wd = m.wave_doubles;
    if (wd < DXO_WAVE * 20) wd = DXO_WAVE * 20;
    wd = (wd + 1) & ~1;
    const size_t shm = (size_t)(m.table_doubles + 4 * wd) * sizeof(double);
    if (shm > 64 * 1024) return dxo_fail(ctx, DXO_E_SIZE, "dxo_isihara_field: element too large for the LDS budget");
    const int64_t n_groups = (n_cells + m.cells_per_wave - 1) / m.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;
    if (ctx->nontemporal != 0) hipLaunchKernelGGL(isihara_field<true>, dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.prm, m, wd, cell0, n_cells, L.d_u, dP, P);
    else hipLaunchKernelGGL(isihara_field<false>, dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.prm, m, wd, cell0, n_cells, L.d_u, dP, P);
    return DXO_OK;
}

int isi_field_chunk(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const*, void* const* d_out, hipStream_t s) {
    IsiFieldLaunch& L = *static_cast<IsiFieldLaunch*>(user);
    const int64_t cell0 = L.next_cell;
    L.next_cell += n_chunk;
    return isi_field_launch(ctx, L, cell0, n_chunk, (double*)d_out[0], (double*)d_out[1], s);
}

int upload_u(dxo_ctx* ctx, dxo_mesh* mesh, const double* u) {
    const size_t ub = (size_t)mesh->num_field_nodes * mesh->gdim * sizeof(double);
    if (mesh->u_cap < ub) {
        if (mesh->d_u) DXO_HIP(ctx, hipFree(mesh->d_u));
        mesh->d_u = nullptr;
        mesh->u_cap = 0;
        DXO_HIP(ctx, hipMalloc((void**)&mesh->d_u, ub));
        mesh->u_cap = ub;
    }
    DXO_HIP(ctx, hipMemcpy(mesh->d_u, u, ub, hipMemcpyHostToDevice));
    return DXO_OK;
}

struct FieldOp {
    dxo_mesh* mesh;
    const double* d_u;
    int kind;               // DXO_OPERAND_EPS_MANDEL or DXO_OPERAND_DEFGRAD
    int64_t next_cell = 0;
    // the consumer: (ctx, staged operand, n points, chunk inputs, chunk outputs, stream)
    std::function<int(dxo_ctx*, const double*, int64_t, void* const*, void* const*, hipStream_t)> consume;
};

int field_op_launch(dxo_ctx* ctx, FieldOp& L, int64_t cell0, int64_t n_cells, void* const* d_in, void* const* d_out, hipStream_t s) {
    if (n_cells == 0) return DXO_OK;
    const int nq = L.mesh->dev.nq;
    const int64_t n = n_cells * nq;
    double* stage = static_cast<double*>(dxo_stage(ctx, s, (size_t)n * 4 * sizeof(double)));
    if (!stage) return dxo_hip_fail(ctx, hipErrorOutOfMemory, "field operator: staging buffer");
    int rc = dxo_operand_launch_range(ctx, L.mesh, L.kind, L.mesh->gdim, L.d_u, cell0, n_cells, stage, s);
    if (rc != DXO_OK) return dxo_fail(ctx, rc, "field operator: operand kernel");
    return L.consume(ctx, stage, n, d_in, d_out, s);
}

int field_op_chunk(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const* d_in, void* const* d_out, hipStream_t s) {
    FieldOp& L = *static_cast<FieldOp*>(user);
    const int64_t cell0 = L.next_cell;
    L.next_cell += n_chunk;
    return field_op_launch(ctx, L, cell0, n_chunk, d_in, d_out, s);
}

// Common driver: device path = two launches over all cells; host path = upload u, then the chunked pipeline over cells.
int run_field_op(dxo_ctx* ctx, FieldOp& L, int mem, const double* u, const std::vector<dxo_span>& in, const std::vector<dxo_span>& out,
                 void* const* dev_in, void* const* dev_out) {
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t nc = L.mesh->num_cells;
    if (mem == DXO_MEM_DEVICE) {
        L.d_u = u;
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = field_op_launch(ctx, L, 0, nc, dev_in, dev_out, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    int rc = upload_u(ctx, L.mesh, u);
    if (rc != DXO_OK) return rc;
    L.d_u = L.mesh->d_u;
    return dxo_run_host_pipeline(ctx, nc, in, out, field_op_chunk, &L, L.mesh->dev.nq);
}

const char* check_mesh_2d(const dxo_mesh* mesh) {
    if (!mesh) return "mesh is NULL";
    if (mesh->gdim != 2) return "the operator is plane strain / 2-D (Mandel length 4, F 2x2): the mesh must have gdim = 2";
    return nullptr;
}

}  // namespace

extern "C" int dxo_mohr_coulomb_field(dxo_ctx* ctx, const dxo_mc_params* prm, dxo_mesh* mesh, int mem, const double* u,
                                      const double* sigma_n, double* C_tang, double* sigma, int32_t* niter, double* yielding,
                                      double* norm_res, double* dlambda) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb_field: params is NULL");
    if (const char* why = check_mesh_2d(mesh)) return dxo_fail(ctx, mesh ? DXO_E_DIM : DXO_E_NULL, why);
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_mohr_coulomb_field: bad mem");
    if (prm->nitermax < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mohr_coulomb_field: nitermax < 0");
    if (mesh->num_cells == 0) return DXO_OK;
    if (!u || !sigma_n || !C_tang || !sigma) return dxo_fail(ctx, DXO_E_NULL, "dxo_mohr_coulomb_field: NULL array");
    const uintptr_t a16 = (uintptr_t)sigma_n | (uintptr_t)C_tang | (uintptr_t)sigma;
    const uintptr_t a8 = (uintptr_t)u | (uintptr_t)yielding | (uintptr_t)norm_res | (uintptr_t)dlambda;
    if (mem == DXO_MEM_DEVICE && (a16 & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb_field: device sigma_n, C_tang, sigma must be 16-byte aligned");
    if ((a16 & 7u) || (a8 & 7u) || ((uintptr_t)niter & 3u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_mohr_coulomb_field: misaligned array");
    const bool has[4] = {niter != nullptr, yielding != nullptr, norm_res != nullptr, dlambda != nullptr};
    const dxo_mc_params P = *prm;
    FieldOp L{mesh, nullptr, DXO_OPERAND_EPS_MANDEL};
    L.consume = [P, has](dxo_ctx* c, const double* deps, int64_t n, void* const* d_in, void* const* d_out, hipStream_t s) {
        int o = 2;
        int32_t* it = has[0] ? (int32_t*)d_out[o++] : nullptr;
        double* yl = has[1] ? (double*)d_out[o++] : nullptr;
        double* nr = has[2] ? (double*)d_out[o++] : nullptr;
        double* dl = has[3] ? (double*)d_out[o++] : nullptr;
        return dxo_mc_launch_device(c, &P, n, deps, (const double*)d_in[0], (double*)d_out[0], (double*)d_out[1], it, yl, nr, dl, s);
    };
    const size_t sd = sizeof(double) * (size_t)mesh->dev.nq;
    std::vector<dxo_span> in = {{sigma_n, nullptr, 4 * sd}};
    std::vector<dxo_span> out = {{nullptr, C_tang, 16 * sd}, {nullptr, sigma, 4 * sd}};
    std::vector<void*> dout = {C_tang, sigma};
    if (niter) { out.push_back({nullptr, niter, sizeof(int32_t) * (size_t)mesh->dev.nq}); dout.push_back(niter); }
    if (yielding) { out.push_back({nullptr, yielding, sd}); dout.push_back(yielding); }
    if (norm_res) { out.push_back({nullptr, norm_res, sd}); dout.push_back(norm_res); }
    if (dlambda) { out.push_back({nullptr, dlambda, sd}); dout.push_back(dlambda); }
    void* din[1] = {const_cast<double*>(sigma_n)};
    return run_field_op(ctx, L, mem, u, in, out, din, dout.data());
}

extern "C" int dxo_icnn_field(dxo_ctx* ctx, const dxo_icnn* model, int precision, dxo_mesh* mesh, int mem, const double* u,
                              double* dP, double* P) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!model) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_field: model is NULL");
    if (const char* why = check_mesh_2d(mesh)) return dxo_fail(ctx, mesh ? DXO_E_DIM : DXO_E_NULL, why);
    if (precision != 0 && precision != 1) return dxo_fail(ctx, DXO_E_OPTION, "dxo_icnn_field: precision must be 0 (fp32 network) or 1 (fp64)");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_icnn_field: bad mem");
    if (mesh->num_cells == 0) return DXO_OK;
    if (!u || !dP || !P) return dxo_fail(ctx, DXO_E_NULL, "dxo_icnn_field: NULL array");
    const uintptr_t all = (uintptr_t)u | (uintptr_t)dP | (uintptr_t)P;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_icnn_field: arrays must be 8-byte aligned");
    if (mem == DXO_MEM_DEVICE && (((uintptr_t)dP | (uintptr_t)P) & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_icnn_field: device dP, P must be 16-byte aligned");
    FieldOp L{mesh, nullptr, DXO_OPERAND_DEFGRAD};
    L.consume = [model, precision](dxo_ctx* c, const double* F, int64_t n, void* const*, void* const* d_out, hipStream_t s) {
        return dxo_icnn_launch_device(c, model, precision, n, F, (double*)d_out[0], (double*)d_out[1], s);
    };
    const size_t sd = sizeof(double) * (size_t)mesh->dev.nq;
    std::vector<dxo_span> in;
    std::vector<dxo_span> out = {{nullptr, dP, 16 * sd}, {nullptr, P, 4 * sd}};
    void* dout[2] = {dP, P};
    return run_field_op(ctx, L, mem, u, in, out, nullptr, dout);
}

extern "C" int dxo_isihara_field(dxo_ctx* ctx, const dxo_isihara_params* prm, dxo_mesh* mesh, int mem, const double* u,
                                 double* dP, double* P) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm) return dxo_fail(ctx, DXO_E_NULL, "dxo_isihara_field: params is NULL");
    if (const char* why = check_mesh_2d(mesh)) return dxo_fail(ctx, mesh ? DXO_E_DIM : DXO_E_NULL, why);
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_isihara_field: bad mem");
    if (mesh->num_cells == 0) return DXO_OK;
    if (!u || !dP || !P) return dxo_fail(ctx, DXO_E_NULL, "dxo_isihara_field: NULL array");
    const uintptr_t all = (uintptr_t)u | (uintptr_t)dP | (uintptr_t)P;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_isihara_field: arrays must be 8-byte aligned");
    if (mem == DXO_MEM_DEVICE && (((uintptr_t)dP | (uintptr_t)P) & 15u)) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_isihara_field: device dP, P must be 16-byte aligned");
    IsiFieldLaunch L{IsiPrm{prm->c1, prm->c2, prm->c3, prm->c4}, mesh, u, 0};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (mem == DXO_MEM_DEVICE) {
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = isi_field_launch(ctx, L, 0, mesh->num_cells, dP, P, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    int rc = upload_u(ctx, mesh, u);
    if (rc != DXO_OK) return rc;
    L.d_u = mesh->d_u;
    const size_t sd = sizeof(double) * (size_t)mesh->dev.nq;
    std::vector<dxo_span> in;
    std::vector<dxo_span> out = {{nullptr, dP, 16 * sd}, {nullptr, P, 4 * sd}};
    return dxo_run_host_pipeline(ctx, mesh->num_cells, in, out, isi_field_chunk, &L, mesh->dev.nq, nullptr, true);
}
