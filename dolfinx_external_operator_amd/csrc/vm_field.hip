// vm_field.hip — operand evaluation fused in front of the von Mises return map.
//
// The reference's residual evaluation does, per SNES iteration (demo_plasticity_von_mises.py:445-456):
//   evaluated_operands = evaluate_operands(F_external_operators)          # deps = eps(Du) at every quadrature point
//   ((_, sigma_new, dp_new),) = evaluate_external_operators(J_external_operators, evaluated_operands)
// i.e. Expression.eval writes deps (num_cells, nq, d) and the constitutive kernel reads it back. Here one launch
// does both: the wave gathers its cells' displacement dofs, each lane forms the strain increment of its own
// quadrature point in registers (operand_core.h) and runs the radial return on it (vm_core.h); deps never exists
// in memory. Traffic per point at d = 6 on Q2 hexahedra: read sigma_n 48 + p 8 + ~45 B of dofs/dofmaps,
// write 344 B — against 93 B + 448 B for the two separate launches.
//
// Tiling: a wave owns floor(64 / nq) consecutive cells = up to 64 consecutive points (63 for nq = 3); sigma_n,
// sigma and C_tang move exactly as in vm_tile (lane-linear 16-byte accesses through wave-private LDS, tangent
// rebuilt in output order). The operand's gather buffer and vm_tile's staging slices share the wave's LDS region.
#include "dxo_common.h"
#include "operand_core.h"
#include "vm_core.h"

namespace {

#ifndef DXO_VMF_WAVES
#define DXO_VMF_WAVES 3
#endif
#ifndef DXO_OP_CT
#define DXO_OP_CT 1
#endif
template <int G, bool NT, int ND_CT = 0, int NG_CT = 0>
__global__ __launch_bounds__(DXO_BLOCK, DXO_VMF_WAVES) void vm_field(VmConst c, OperandDev m, int wave_doubles, int64_t cell0,
                                                         int64_t n_cells, const double* __restrict__ u,
                                                         const double* __restrict__ sigma_n,
                                                         const double* __restrict__ p, double* __restrict__ C_tang,
                                                         double* __restrict__ sigma, double* __restrict__ dp_out) {
    constexpr int D = G == 2 ? 4 : 6;
    using T = VmTile<D>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* tab = lds;
    operand_load_tables<G>(m, tab);
    __syncthreads();
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* W = lds + m.table_doubles + wave * wave_doubles;   // operand gather buffer, then X | Y of vm_tile
    double* X = W;
    double* Y = X + T::X_DOUBLES;
    dxo_f64x2* X2 = reinterpret_cast<dxo_f64x2*>(X);
    dxo_f64x2* Y2 = reinterpret_cast<dxo_f64x2*>(Y);
    const int cpw = m.cells_per_wave;
    const int64_t n_groups = (n_cells + cpw - 1) / cpw;
    const GroupWalk walk = xcd_group_walk(n_groups, T::WAVES, wave);
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    // the dof gather runs as a register pipeline two groups ahead (operand_core.h) whenever the element fits it
    const bool piped = operand_can_pipe(m);
    OperandPipe<G, G> pf;
    int64_t grp = walk.first;
    if (piped) {
        pipe_load_indices<G, G>(m, pf, cell0 + grp * cpw, cells_in(grp), lane);
        pipe_load_values<G, G>(m, pf, u);
        pipe_load_indices<G, G>(m, pf, cell0 + (grp + stride) * cpw, cells_in(grp + stride), lane);
    }
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;                 // first cell of the group, relative to cell0
        const int ncell = cells_in(grp);
        const int npts = ncell * m.nq;
        const int64_t p0 = c0 * m.nq;                 // first point, relative to the state/output arrays passed in
        const int nvec = npts * T::CH_VEC;

        // ---- A1: strain increment of this lane's point from the displacement dofs
        double e[D];
        bool active;
        if (piped) {
            pipe_commit<G, G>(m, pf, W, ncell, lane);
            pipe_load_values<G, G>(m, pf, u);
            pipe_load_indices<G, G>(m, pf, cell0 + (grp + 2 * stride) * cpw, cells_in(grp + 2 * stride), lane);
            active = operand_compute<G, G, DXO_OPERAND_EPS_MANDEL, ND_CT, NG_CT>(m, tab, W, ncell, lane, e);
        } else {
            active = operand_point<G, G, DXO_OPERAND_EPS_MANDEL>(m, tab, W, u, nullptr, cell0 + c0, ncell, lane, e);
        }
        if (!active) {
#pragma unroll
            for (int k = 0; k < D; ++k) e[k] = 0.0;
        }
        // ---- A2: sigma_n lane-linear -> LDS -> point-per-lane
        const dxo_f64x2* g_s = reinterpret_cast<const dxo_f64x2*>(sigma_n + p0 * D);
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const int idx = k * DXO_WAVE + lane;
            Y2[idx] = idx < nvec ? g_s[idx] : dxo_f64x2{0.0, 0.0};
        }
        const double p_l = lane < npts ? p[p0 + lane] : 0.0;
        wave_lds_fence();
        double sn[D];
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const dxo_f64x2 b2 = Y2[lane * T::CH_VEC + k];
            sn[2 * k] = b2.x;
            sn[2 * k + 1] = b2.y;
        }
        wave_lds_fence();

        // ---- B: radial return
        double sig[D], nrm[D], dp, a, b;
        vm_return_map<D>(c, e, sn, p_l, sig, dp, nrm, a, b);

        // ---- C: as vm_tile
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            X2[lane * T::CH_VEC + k] = dxo_f64x2{sig[2 * k], sig[2 * k + 1]};
            Y2[lane * (T::ST / 2) + k] = dxo_f64x2{nrm[2 * k], nrm[2 * k + 1]};
        }
        Y2[lane * (T::ST / 2) + T::CH_VEC] = dxo_f64x2{a, b};
        wave_lds_fence();
        if (lane < npts) store8<NT>(dp_out + p0 + lane, dp);
        dxo_f64x2* g_o = reinterpret_cast<dxo_f64x2*>(sigma + p0 * D);
#pragma unroll
        for (int k = 0; k < T::CH_VEC; ++k) {
            const int idx = k * DXO_WAVE + lane;
            if (idx < nvec) store16<NT>(g_o + idx, X2[idx]);
        }
        vm_store_tangent<D, NT>(c, Y, reinterpret_cast<dxo_f64x2*>(C_tang + p0 * (D * D)), npts * T::CH_CT, lane);
        wave_lds_fence();
    }
}

struct FieldLaunch {
    VmConst c;
    const dxo_mesh* mesh;
    const double* d_u;
    int64_t next_cell;     // host pipeline: chunks arrive in order, each covers the next n_chunk cells
};

int field_launch(dxo_ctx* ctx, const FieldLaunch& L, int64_t cell0, int64_t n_cells, const double* sigma_n,
                 const double* p, double* C_tang, double* sigma, double* dp, hipStream_t s) {
    if (n_cells == 0) return DXO_OK;
    const OperandDev& m = L.mesh->dev;
    const int D = L.mesh->gdim == 2 ? 4 : 6;
    int wd = m.cells_per_wave * (op_odd(m.ndofs * L.mesh->gdim) + op_odd(m.ngeom * L.mesh->gdim));
    const int tile = DXO_WAVE * D + DXO_WAVE * (D + 2);
    if (wd < tile) wd = tile;
    wd = (wd + 1) & ~1;
    const size_t shm = (size_t)(m.table_doubles + 4 * wd) * sizeof(double);
    if (shm > 64 * 1024) return dxo_fail(ctx, DXO_E_SIZE, "dxo_von_mises_field: element too large for the LDS budget");
    const int64_t n_groups = (n_cells + m.cells_per_wave - 1) / m.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 16;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;      // whole rounds over the 8 XCDs (xcd_group_walk)
    const bool nt = ctx->nontemporal != 0;
    if (DXO_OP_CT && L.mesh->gdim == 3 && m.ndofs == 27 && m.ngeom == 8) {   // Q2 hexahedra: trip counts known at compile time
        if (nt) hipLaunchKernelGGL((vm_field<3, true, 27, 8>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
        else    hipLaunchKernelGGL((vm_field<3, false, 27, 8>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
        return DXO_OK;
    }
    if (L.mesh->gdim == 2) {
        if (nt) hipLaunchKernelGGL((vm_field<2, true>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
        else    hipLaunchKernelGGL((vm_field<2, false>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
    } else {
        if (nt) hipLaunchKernelGGL((vm_field<3, true>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
        else    hipLaunchKernelGGL((vm_field<3, false>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, L.c, m, wd, cell0, n_cells, L.d_u, sigma_n, p, C_tang, sigma, dp);
    }
    return DXO_OK;
}

int field_chunk(dxo_ctx* ctx, void* user, int64_t n_chunk, void* const* d_in, void* const* d_out, hipStream_t s) {
    FieldLaunch& L = *static_cast<FieldLaunch*>(user);
    const int64_t cell0 = L.next_cell;
    L.next_cell += n_chunk;
    return field_launch(ctx, L, cell0, n_chunk, (const double*)d_in[0], (const double*)d_in[1], (double*)d_out[0],
                        (double*)d_out[1], (double*)d_out[2], s);
}

}  // namespace

extern "C" int dxo_von_mises_field(dxo_ctx* ctx, const dxo_vm_params* prm, dxo_mesh* mesh, int mem, const double* u,
                                   const double* sigma_n, const double* p, double* C_tang, double* sigma, double* dp) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!prm || !mesh) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field: NULL params or mesh");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_von_mises_field: bad mem");
    const int64_t nc = mesh->num_cells;
    if (nc == 0) return DXO_OK;
    if (!u || !sigma_n || !p || !C_tang || !sigma || !dp) return dxo_fail(ctx, DXO_E_NULL, "dxo_von_mises_field: NULL array");
    const uintptr_t all = (uintptr_t)u | (uintptr_t)sigma_n | (uintptr_t)p | (uintptr_t)C_tang | (uintptr_t)sigma | (uintptr_t)dp;
    if (all & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field: arrays must be 8-byte aligned");
    const int G = mesh->gdim, D = G == 2 ? 4 : 6, nq = mesh->dev.nq;
    FieldLaunch L{make_const(*prm), mesh, u, 0};
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (mem == DXO_MEM_DEVICE) {
        if (((uintptr_t)sigma_n | (uintptr_t)C_tang | (uintptr_t)sigma) & 15u)
            return dxo_fail(ctx, DXO_E_ALIGN, "dxo_von_mises_field: device sigma_n, C_tang, sigma must be 16-byte aligned");
        hipStream_t s = dxo_launch_stream(ctx);
        int rc = dxo_device_begin(ctx, s);
        if (rc != DXO_OK) return rc;
        rc = field_launch(ctx, L, 0, nc, sigma_n, p, C_tang, sigma, dp, s);
        if (rc != DXO_OK) return rc;
        return dxo_device_end(ctx, s);
    }
    // host arrays: the field vector goes up whole (it is small: one value per dof, not per quadrature point), the
    // state and the outputs stream through the chunked pipeline in units of CELLS
    const size_t ub = (size_t)mesh->num_field_nodes * G * sizeof(double);
    if (mesh->u_cap < ub) {
        if (mesh->d_u) DXO_HIP(ctx, hipFree(mesh->d_u));
        mesh->d_u = nullptr;
        mesh->u_cap = 0;
        DXO_HIP(ctx, hipMalloc((void**)&mesh->d_u, ub));
        mesh->u_cap = ub;
    }
    DXO_HIP(ctx, hipMemcpy(mesh->d_u, u, ub, hipMemcpyHostToDevice));
    L.d_u = mesh->d_u;
    const size_t sd = sizeof(double) * (size_t)nq;
    std::vector<dxo_span> in = {{sigma_n, nullptr, D * sd}, {p, nullptr, sd}};
    std::vector<dxo_span> out = {{nullptr, C_tang, D * D * sd}, {nullptr, sigma, D * sd}, {nullptr, dp, sd}};
    return dxo_run_host_pipeline(ctx, nc, in, out, field_chunk, &L, nq);
}
