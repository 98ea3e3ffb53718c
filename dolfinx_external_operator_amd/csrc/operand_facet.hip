// operand_facet.hip — operand evaluation on codim-1 entities: `(cell, local facet)` pairs.
//
// Reference: evaluate_operands passes its `entities` argument straight to `fem.Expression.eval`
// (src/dolfinx_external_operator/external_operator.py:340, 402); for an operator living on a facet sub-mesh the
// entities are (cell, local_facet) pairs (test/test_codim_external_operator.py:76-84, 111) and the Expression's
// points are quadrature points of the reference FACET, which DOLFINx maps into the reference cell per local facet
// before tabulating. The device counterpart therefore takes the tables tabulated at those mapped points, one set per
// local facet of the cell (dxo_mesh_set_facet_tables: basix `tabulate(1, facet_points_in_cell[f])` for f = 0..nf-1),
// and evaluates value / gradient / eps / F at the nq_f points of every entity; the cell's full Jacobian is rebuilt at
// each point, so the gradient is the physical gradient of the field (not a tangential one), as Expression.eval gives.
// Output layout: (n_entities, nq_f, value_size), C order.
//
// Boundary integrals touch O(N^((d-1)/d)) points — this is not a bandwidth kernel: one lane per (entity, point),
// dofs gathered through the dofmap and tables read straight from global memory (they stay in L1 / L2).
#include "dxo_common.h"
#include "operand_core.h"

namespace {

template <int G, int BS, int KIND>
__global__ __launch_bounds__(DXO_BLOCK) void operand_eval_facets(OperandDev m, int nqf, const double* __restrict__ phi_f,
                                                                 const double* __restrict__ dphi_f,
                                                                 const double* __restrict__ dpsi_f,
                                                                 const double* __restrict__ u,
                                                                 const int32_t* __restrict__ ents, int64_t n_ents,
                                                                 double* __restrict__ out) {
    constexpr int D = OperandShape<G, BS, KIND>::D;
    const int nd = m.ndofs, ng = m.ngeom;
    const int64_t total = n_ents * nqf, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t e = t / nqf;
        const int q = (int)(t - e * nqf);
        const int64_t cell = ents[2 * e];
        const int f = ents[2 * e + 1];
        const double* phi = phi_f + ((size_t)f * nqf + q) * nd;
        const double* dphi = dphi_f + ((size_t)f * nqf + q) * nd * G;
        const double* dpsi = dpsi_f + ((size_t)f * nqf + q) * ng * G;
        double J[G][G], K[G][G];
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int k = 0; k < G; ++k) J[j][k] = 0.0;
        for (int v = 0; v < ng; ++v) {
            const int64_t node = m.geom_dofmap[cell * ng + v];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const double xj = m.x[node * G + j];
#pragma unroll
                for (int k = 0; k < G; ++k) J[j][k] += xj * dpsi[v * G + k];
            }
        }
        (void)invert<G>(J, K);
        double val[BS], gref[BS][G];
#pragma unroll
        for (int i = 0; i < BS; ++i) {
            val[i] = 0.0;
#pragma unroll
            for (int k = 0; k < G; ++k) gref[i][k] = 0.0;
        }
        for (int a = 0; a < nd; ++a) {
            const int64_t node = m.dofmap[cell * nd + a];
            const double ph = phi[a];
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                const double ua = u[node * (m.u_stride ? m.u_stride : BS) + i];
                val[i] += ua * ph;
#pragma unroll
                for (int k = 0; k < G; ++k) gref[i][k] += ua * dphi[a * G + k];
            }
        }
        double g[BS][G];
#pragma unroll
        for (int i = 0; i < BS; ++i)
#pragma unroll
            for (int j = 0; j < G; ++j) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < G; ++k) s += gref[i][k] * K[k][j];
                g[i][j] = s;
            }
        double o[D];
        shape_operand<G, BS, KIND>(val, g, o);
#pragma unroll
        for (int k = 0; k < D; ++k) out[t * (m.out_stride ? m.out_stride : D) + k] = o[k];
    }
}

template <int G, int BS, int KIND>
void launch_facets(const dxo_ctx* ctx, const dxo_mesh* m, const double* u, const int32_t* ents, int64_t n, double* out, hipStream_t s,
                   int u_stride = 0, int out_stride = 0) {
    const size_t nd = (size_t)m->dev.ndofs, nf = (size_t)m->n_local_facets, nqf = (size_t)m->nq_facet;
    const double* phi_f = m->d_facet_tab;
    const double* dphi_f = phi_f + nf * nqf * nd;
    const double* dpsi_f = dphi_f + nf * nqf * nd * G;
    int64_t blocks = (n * (int64_t)nqf + DXO_BLOCK - 1) / DXO_BLOCK;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    OperandDev dev = m->dev;
    dev.u_stride = u_stride;
    dev.out_stride = out_stride;
    hipLaunchKernelGGL((operand_eval_facets<G, BS, KIND>), dim3((int)blocks), dim3(DXO_BLOCK), 0, s, dev, (int)nqf, phi_f, dphi_f,
                       dpsi_f, u, ents, n, out);
}

// any block size for the per-component kinds, as dispatch_components of operand.hip: one scalar launch per component
template <int G>
int dispatch_facet_components(const dxo_ctx* ctx, const dxo_mesh* m, int kind, int bs, const double* u, const int32_t* ents, int64_t n,
                              double* out, hipStream_t s) {
    const int os = kind == DXO_OPERAND_VALUE ? bs : kind == DXO_OPERAND_GRAD ? bs * G : bs * (1 + G);
    for (int c = 0; c < bs; ++c) {
        if (kind == DXO_OPERAND_VALUE || kind == DXO_OPERAND_VALUE_GRAD)
            launch_facets<G, 1, DXO_OPERAND_VALUE>(ctx, m, u + c, ents, n, out + c, s, bs, os);
        if (kind == DXO_OPERAND_GRAD || kind == DXO_OPERAND_VALUE_GRAD)
            launch_facets<G, 1, DXO_OPERAND_GRAD>(ctx, m, u + c, ents, n, out + (kind == DXO_OPERAND_GRAD ? 0 : bs) + c * G, s, bs, os);
    }
    return DXO_OK;
}

template <int G, int BS>
int dispatch_facets(const dxo_ctx* ctx, const dxo_mesh* m, int kind, const double* u, const int32_t* ents, int64_t n, double* out,
                    hipStream_t s) {
    switch (kind) {
        case DXO_OPERAND_VALUE: launch_facets<G, BS, DXO_OPERAND_VALUE>(ctx, m, u, ents, n, out, s); return DXO_OK;
        case DXO_OPERAND_GRAD: launch_facets<G, BS, DXO_OPERAND_GRAD>(ctx, m, u, ents, n, out, s); return DXO_OK;
        case DXO_OPERAND_VALUE_GRAD: launch_facets<G, BS, DXO_OPERAND_VALUE_GRAD>(ctx, m, u, ents, n, out, s); return DXO_OK;
        case DXO_OPERAND_EPS_MANDEL:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_EPS_MANDEL>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DEFGRAD:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_DEFGRAD>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_CAUCHY_GREEN:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_CAUCHY_GREEN>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_I1:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_I1>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DETF:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_DETF>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DIV:
            if constexpr (BS == G) { launch_facets<G, BS, DXO_OPERAND_DIV>(ctx, m, u, ents, n, out, s); return DXO_OK; }
            return DXO_E_DIM;
    }
    return DXO_E_OPTION;
}

int ensure_buf(dxo_ctx* ctx, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return DXO_OK;
    if (*p) DXO_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    DXO_HIP(ctx, hipMalloc(p, bytes));
    *cap = bytes;
    return DXO_OK;
}

}  // namespace

extern "C" int dxo_mesh_set_facet_tables(dxo_ctx* ctx, dxo_mesh* m, int n_local_facets, int nq, const double* phi,
                                         const double* dphi, const double* dpsi) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m || !phi || !dphi || !dpsi) return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_set_facet_tables: NULL argument");
    if (n_local_facets < 1 || n_local_facets > 6 || nq < 1) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_set_facet_tables: 1..6 local facets, nq >= 1");
    const size_t G = (size_t)m->gdim, nf = (size_t)n_local_facets, n_phi = nf * nq * m->dev.ndofs, n_dphi = n_phi * G,
                 n_dpsi = nf * nq * m->dev.ngeom * G;
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    DXO_HIP(ctx, hipDeviceSynchronize());
    if (m->d_facet_tab) DXO_HIP(ctx, hipFree(m->d_facet_tab));
    m->d_facet_tab = nullptr;
    m->n_local_facets = m->nq_facet = 0;
    DXO_HIP(ctx, hipMalloc((void**)&m->d_facet_tab, (n_phi + n_dphi + n_dpsi) * sizeof(double)));
    DXO_HIP(ctx, hipMemcpy(m->d_facet_tab, phi, n_phi * sizeof(double), hipMemcpyHostToDevice));
    DXO_HIP(ctx, hipMemcpy(m->d_facet_tab + n_phi, dphi, n_dphi * sizeof(double), hipMemcpyHostToDevice));
    DXO_HIP(ctx, hipMemcpy(m->d_facet_tab + n_phi + n_dphi, dpsi, n_dpsi * sizeof(double), hipMemcpyHostToDevice));
    m->n_local_facets = n_local_facets;
    m->nq_facet = nq;
    return DXO_OK;
}

extern "C" int dxo_eval_operand_facets(dxo_ctx* ctx, dxo_mesh* m, int kind, int bs, int mem, const double* u,
                                       const int32_t* entities, int64_t n_entities, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_operand_facets: mesh is NULL");
    if (!m->d_facet_tab) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_operand_facets: call dxo_mesh_set_facet_tables first");
    const int D = dxo_operand_value_size(m->gdim, bs, kind);
    if (D == DXO_E_OPTION) return dxo_fail(ctx, DXO_E_OPTION, "dxo_eval_operand_facets: unknown operand kind");
    if (D < 0) return dxo_fail(ctx, DXO_E_DIM, "dxo_eval_operand_facets: block size does not fit the operand kind / gdim");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_eval_operand_facets: bad mem");
    if (n_entities < 0) return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_operand_facets: n_entities < 0");
    if (n_entities == 0) return DXO_OK;
    if (!u || !entities || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_operand_facets: NULL array");
    if (((uintptr_t)u | (uintptr_t)out) & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_eval_operand_facets: arrays must be 8-byte aligned");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const double* du = u;
    const int32_t* de = entities;
    double* dout = out;
    const size_t out_bytes = (size_t)n_entities * m->nq_facet * D * sizeof(double);
    if (mem == DXO_MEM_HOST) {
        for (int64_t i = 0; i < n_entities; ++i)
            if (entities[2 * i] < 0 || entities[2 * i] >= m->num_cells || entities[2 * i + 1] < 0 || entities[2 * i + 1] >= m->n_local_facets)
                return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_operand_facets: entity outside [0, num_cells) x [0, n_local_facets)");
        const size_t ub = (size_t)m->num_field_nodes * bs * sizeof(double);
        int rc = ensure_buf(ctx, (void**)&m->d_u, &m->u_cap, ub);
        if (rc != DXO_OK) return rc;
        DXO_HIP(ctx, hipMemcpyAsync(m->d_u, u, ub, hipMemcpyHostToDevice, s));
        du = m->d_u;
        rc = ensure_buf(ctx, (void**)&m->d_ents, &m->ents_cap, (size_t)n_entities * 2 * sizeof(int32_t));
        if (rc != DXO_OK) return rc;
        DXO_HIP(ctx, hipMemcpyAsync(m->d_ents, entities, (size_t)n_entities * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
        de = m->d_ents;
        rc = ensure_buf(ctx, (void**)&m->d_out, &m->out_cap, out_bytes);
        if (rc != DXO_OK) return rc;
        dout = m->d_out;
    }
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    if (bs != 1 && bs != m->gdim) rc = m->gdim == 2 ? dispatch_facet_components<2>(ctx, m, kind, bs, du, de, n_entities, dout, s) : dispatch_facet_components<3>(ctx, m, kind, bs, du, de, n_entities, dout, s);
    else if (m->gdim == 2) rc = bs == 1 ? dispatch_facets<2, 1>(ctx, m, kind, du, de, n_entities, dout, s) : dispatch_facets<2, 2>(ctx, m, kind, du, de, n_entities, dout, s);
    else              rc = bs == 1 ? dispatch_facets<3, 1>(ctx, m, kind, du, de, n_entities, dout, s) : dispatch_facets<3, 3>(ctx, m, kind, du, de, n_entities, dout, s);
    if (rc != DXO_OK) return dxo_fail(ctx, rc, "dxo_eval_operand_facets: unsupported (gdim, bs, kind)");
    rc = dxo_device_end(ctx, s);
    if (rc != DXO_OK) return rc;
    if (mem == DXO_MEM_HOST) {
        DXO_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));
    }
    return DXO_OK;
}
