// operand.hip — operand evaluation on the device.
//
// Reference: evaluate_operands builds `fem.Expression(operand, quadrature_points)` and calls
// `expr.eval(mesh, entities)` (src/dolfinx_external_operator/external_operator.py:386-402): a DOLFINx/FFCx cell loop
// that tabulates the operand at every quadrature point of every cell. Every operand of the reference's demos is
// a linear function of the gradient (or the value) of one Lagrange field:
//   eps(Du) = [g00, g11, 0, sqrt(1/2)(g01 + g10)]     demo_plasticity_von_mises.py:225-227, _mohr_coulomb.py:148-157
//   F = I + grad u                                     demo_hyperelasticity.py:479
//   T, grad T                                          demo_nonlinear_heat_equation_part2.py (operands of q)
// so the device version is: gather the cell's dofs through the dofmap, contract with the tabulated reference
// gradients at the cell's quadrature points, push forward with J^-1 built from the geometry dofs, shape the
// operand. Output layout is Expression.eval's: (num_cells, nq, value_size), C order.
//
// Kernel shape: lane = (cell, quadrature point); a wave owns floor(64 / nq) consecutive cells. The wave gathers
// its cells' field dofs and geometry into wave-private LDS (the irregular part: int32 dofmap -> 8-byte loads,
// neighbours share nodes so most hits come from L2), the reference tables sit in LDS once per workgroup, each
// lane accumulates Gref = sum_a u_a (x) dphi_a(q), multiplies by J^-1, and the wave writes its points in output
// order through LDS (consecutive 8-byte words per store instruction). No __syncthreads inside the cell loop.
#include "dxo_common.h"
#include "operand_core.h"
#include "operand_cell.h"

namespace {

template <int G, int BS, int KIND>
__global__ __launch_bounds__(DXO_BLOCK) void operand_eval(OperandDev m, const double* __restrict__ u,
                                                          const int32_t* __restrict__ cells, int64_t n_cells,
                                                          double* __restrict__ out) {
    constexpr int D = OperandShape<G, BS, KIND>::D;
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
    const int64_t stride = walk.stride;
    auto cells_in = [&](int64_t g) -> int {
        if (g >= walk.end) return 0;
        const int64_t left = n_cells - g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    auto store_points = [&](int64_t c0, int ncell, bool active, const double (&o)[D]) {
        // output-ordered store: the wave's ncell*nq points are consecutive in `out`
        if (active) {
#pragma unroll
            for (int k = 0; k < D; ++k) W[lane * D + k] = o[k];
        }
        op_fence();
        const int nval = ncell * m.nq * D;
        if (m.out_stride) {      // one component of a wider operand: D values per point, out_stride apart
            double* g_s = out + c0 * m.nq * m.out_stride;
            for (int idx = lane; idx < nval; idx += DXO_WAVE) g_s[(int64_t)(idx / D) * m.out_stride + idx % D] = W[idx];
        } else {
            double* g_o = out + c0 * m.nq * D;
            for (int idx = lane; idx < nval; idx += DXO_WAVE) __builtin_nontemporal_store(W[idx], g_o + idx);
        }
        op_fence();
    };
    int64_t grp = walk.first;
    if (cells == nullptr && operand_can_pipe(m)) {
        OperandPipe<G, BS> pf;
        pipe_load_indices<G, BS>(m, pf, grp * cpw, cells_in(grp), lane);
        pipe_load_values<G, BS>(m, pf, u);
        pipe_load_indices<G, BS>(m, pf, (grp + stride) * cpw, cells_in(grp + stride), lane);
        for (; grp < walk.end; grp += stride) {
            const int ncell = cells_in(grp);
            pipe_commit<G, BS>(m, pf, W, ncell, lane);                       // group g: registers -> LDS
            pipe_load_values<G, BS>(m, pf, u);                               // group g+1 in flight during the compute
            pipe_load_indices<G, BS>(m, pf, (grp + 2 * stride) * cpw, cells_in(grp + 2 * stride), lane);
            double o[D];
            const bool active = operand_compute<G, BS, KIND>(m, tab, W, ncell, lane, o);
            store_points(grp * cpw, ncell, active, o);
        }
        return;
    }
    for (; grp < walk.end; grp += stride) {
        const int64_t c0 = grp * cpw;
        const int ncell = cells_in(grp);
        double o[D];
        const bool active = operand_point<G, BS, KIND>(m, tab, W, u, cells, c0, ncell, lane, o);
        store_points(c0, ncell, active, o);
    }
}

}  // namespace


namespace {

template <int G, int BS, int KIND>
void launch_operand_dev(const dxo_ctx* ctx, const OperandDev& dev, const double* u, const int32_t* cells, int64_t n_cells,
                        double* out, hipStream_t s) {
    const int64_t n_groups = (n_cells + dev.cells_per_wave - 1) / dev.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;      // whole rounds over the 8 XCDs (xcd_group_walk)
    const size_t shm = (size_t)(dev.table_doubles + 4 * dev.wave_doubles) * sizeof(double);
    hipLaunchKernelGGL((operand_eval<G, BS, KIND>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, dev, u, cells, n_cells, out);
}

template <int G, int BS, int KIND>
void launch_operand(const dxo_ctx* ctx, const dxo_mesh* m, const double* u, const int32_t* cells, int64_t n_cells,
                    double* out, hipStream_t s) {
    const int64_t n_groups = (n_cells + m->dev.cells_per_wave - 1) / m->dev.cells_per_wave;
    int64_t blocks = (n_groups + 3) / 4;
    const int64_t cap = (int64_t)ctx->compute_units * 8;
    if (blocks > cap) blocks = cap;
    blocks = (blocks + 7) / 8 * 8;      // whole rounds over the 8 XCDs (xcd_group_walk)
    const size_t shm = (size_t)(m->dev.table_doubles + 4 * m->dev.wave_doubles) * sizeof(double);
    hipLaunchKernelGGL((operand_eval<G, BS, KIND>), dim3((int)blocks), dim3(DXO_BLOCK), shm, s, m->dev, u, cells, n_cells, out);
}

template <int G, int BS>
int dispatch_kind(const dxo_ctx* ctx, const dxo_mesh* m, int kind, const double* u, const int32_t* cells,
                  int64_t n_cells, double* out, hipStream_t s) {
    switch (kind) {
        case DXO_OPERAND_VALUE: launch_operand<G, BS, DXO_OPERAND_VALUE>(ctx, m, u, cells, n_cells, out, s); return DXO_OK;
        case DXO_OPERAND_GRAD: launch_operand<G, BS, DXO_OPERAND_GRAD>(ctx, m, u, cells, n_cells, out, s); return DXO_OK;
        case DXO_OPERAND_VALUE_GRAD: launch_operand<G, BS, DXO_OPERAND_VALUE_GRAD>(ctx, m, u, cells, n_cells, out, s); return DXO_OK;
        case DXO_OPERAND_EPS_MANDEL:
            if constexpr (BS == G) {
                // standard elements, all cells: lane = cell kernel (registers + scalar tables, no LDS in the contraction)
                if (!cells && ctx->operand_cell && launch_operand_cell_eps(ctx, m, u, n_cells, out, s)) return DXO_OK;
                launch_operand<G, BS, DXO_OPERAND_EPS_MANDEL>(ctx, m, u, cells, n_cells, out, s);
                return DXO_OK;
            }
            return DXO_E_DIM;
        case DXO_OPERAND_DEFGRAD:
            if constexpr (BS == G) { launch_operand<G, BS, DXO_OPERAND_DEFGRAD>(ctx, m, u, cells, n_cells, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_CAUCHY_GREEN:
            if constexpr (BS == G) { launch_operand<G, BS, DXO_OPERAND_CAUCHY_GREEN>(ctx, m, u, cells, n_cells, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_I1:
            if constexpr (BS == G) { launch_operand<G, BS, DXO_OPERAND_I1>(ctx, m, u, cells, n_cells, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DETF:
            if constexpr (BS == G) { launch_operand<G, BS, DXO_OPERAND_DETF>(ctx, m, u, cells, n_cells, out, s); return DXO_OK; }
            return DXO_E_DIM;
        case DXO_OPERAND_DIV:
            if constexpr (BS == G) { launch_operand<G, BS, DXO_OPERAND_DIV>(ctx, m, u, cells, n_cells, out, s); return DXO_OK; }
            return DXO_E_DIM;
    }
    return DXO_E_OPTION;
}

// A Lagrange field of ANY block size (the reference evaluates whatever `fem.Expression` is handed: test/test_nested_ex_op.py:113-118 uses a
// 4-component DG field as an operand) for the kinds that act on each component alone — value, grad, value_grad: one scalar launch per
// component (two for value_grad: its values and its gradients are not adjacent in the operand), reading u at stride bs and writing its slice of
// every point. bs = 1 and bs = gdim take the dense kernels above.
template <int G>
int dispatch_components(const dxo_ctx* ctx, const dxo_mesh* m, int kind, int bs, const double* u, const int32_t* cells, int64_t n_cells,
                        double* out, hipStream_t s) {
    OperandDev dev = m->dev;
    dev.u_stride = bs;
    dev.out_stride = kind == DXO_OPERAND_VALUE ? bs : kind == DXO_OPERAND_GRAD ? bs * G : bs * (1 + G);
    for (int c = 0; c < bs; ++c) {
        if (kind == DXO_OPERAND_VALUE || kind == DXO_OPERAND_VALUE_GRAD)
            launch_operand_dev<G, 1, DXO_OPERAND_VALUE>(ctx, dev, u + c, cells, n_cells, out + c, s);
        if (kind == DXO_OPERAND_GRAD || kind == DXO_OPERAND_VALUE_GRAD)
            launch_operand_dev<G, 1, DXO_OPERAND_GRAD>(ctx, dev, u + c, cells, n_cells, out + (kind == DXO_OPERAND_GRAD ? 0 : bs) + c * G, s);
    }
    return DXO_OK;
}

int ensure(dxo_ctx* ctx, void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return DXO_OK;
    if (*p) DXO_HIP(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    DXO_HIP(ctx, hipMalloc(p, bytes));
    *cap = bytes;
    return DXO_OK;
}

}  // namespace

// Cells [cell0, cell0 + n_cells) of the mesh into out_dev[0 .. n_cells*nq*D): the mesh's dofmaps are simply entered at cell0
// (the kernel sees a mesh of n_cells cells), wave-group kernel. Used by the *_field entry points of field_ops.hip.
int dxo_operand_launch_range(dxo_ctx* ctx, const dxo_mesh* mesh, int kind, int bs, const double* u_dev, int64_t cell0,
                             int64_t n_cells, double* out_dev, hipStream_t s) {
    if (n_cells <= 0) return DXO_OK;
    OperandDev dev = mesh->dev;
    dev.dofmap += cell0 * dev.ndofs;
    dev.geom_dofmap += cell0 * dev.ngeom;
    const int G = mesh->gdim;
    if (bs != G) return DXO_E_DIM;
    if (kind == DXO_OPERAND_EPS_MANDEL) {
        if (G == 2) launch_operand_dev<2, 2, DXO_OPERAND_EPS_MANDEL>(ctx, dev, u_dev, nullptr, n_cells, out_dev, s);
        else launch_operand_dev<3, 3, DXO_OPERAND_EPS_MANDEL>(ctx, dev, u_dev, nullptr, n_cells, out_dev, s);
        return DXO_OK;
    }
    if (kind == DXO_OPERAND_DEFGRAD) {
        if (G == 2) launch_operand_dev<2, 2, DXO_OPERAND_DEFGRAD>(ctx, dev, u_dev, nullptr, n_cells, out_dev, s);
        else launch_operand_dev<3, 3, DXO_OPERAND_DEFGRAD>(ctx, dev, u_dev, nullptr, n_cells, out_dev, s);
        return DXO_OK;
    }
    return DXO_E_OPTION;
}

extern "C" int dxo_operand_value_size(int gdim, int bs, int kind) {
    if (gdim != 2 && gdim != 3) return DXO_E_DIM;
    switch (kind) {
        case DXO_OPERAND_VALUE: return bs >= 1 && bs <= DXO_OPERAND_MAX_BS ? bs : DXO_E_DIM;                      // any block size: per-component kinds
        case DXO_OPERAND_GRAD: return bs >= 1 && bs <= DXO_OPERAND_MAX_BS ? bs * gdim : DXO_E_DIM;
        case DXO_OPERAND_VALUE_GRAD: return bs >= 1 && bs <= DXO_OPERAND_MAX_BS ? bs * (1 + gdim) : DXO_E_DIM;
        case DXO_OPERAND_EPS_MANDEL: return bs == gdim ? (gdim == 2 ? 4 : 6) : DXO_E_DIM;
        case DXO_OPERAND_DEFGRAD: return bs == gdim ? gdim * gdim : DXO_E_DIM;
        case DXO_OPERAND_CAUCHY_GREEN: return bs == gdim ? gdim * gdim : DXO_E_DIM;
        case DXO_OPERAND_I1: return bs == gdim ? 1 : DXO_E_DIM;
        case DXO_OPERAND_DETF: return bs == gdim ? 1 : DXO_E_DIM;
        case DXO_OPERAND_DIV: return bs == gdim ? 1 : DXO_E_DIM;
    }
    return DXO_E_OPTION;
}

extern "C" int dxo_mesh_create(dxo_ctx* ctx, const dxo_mesh_desc* d, dxo_mesh** out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!d || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_create: NULL argument");
    *out = nullptr;
    if (d->gdim != 2 && d->gdim != 3) return dxo_fail(ctx, DXO_E_DIM, "dxo_mesh_create: gdim must be 2 or 3");
    if (d->nq < 1 || d->nq > DXO_WAVE) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_create: 1 <= nq <= 64 required");
    if (d->ndofs < 1 || d->ngeom < d->gdim + 1 || d->num_cells < 0 || d->num_field_nodes < 0 || d->num_geom_nodes < 0)
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_create: bad sizes");
    if (!d->phi || !d->dphi || !d->dpsi || (d->num_cells > 0 && (!d->dofmap || !d->geom_dofmap || !d->x)))
        return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_create: NULL array");
    const int G = d->gdim;
    for (int64_t i = 0; i < d->num_cells * d->ndofs; ++i)
        if (d->dofmap[i] < 0 || d->dofmap[i] >= d->num_field_nodes)
            return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_create: dofmap entry outside [0, num_field_nodes)");
    for (int64_t i = 0; i < d->num_cells * d->ngeom; ++i)
        if (d->geom_dofmap[i] < 0 || d->geom_dofmap[i] >= d->num_geom_nodes)
            return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_create: geometry dofmap entry outside [0, num_geom_nodes)");
    dxo_mesh* m = new dxo_mesh();
    m->gdim = G;
    m->num_cells = d->num_cells;
    m->num_field_nodes = d->num_field_nodes;
    m->num_geom_nodes = d->num_geom_nodes;
    OperandDev& v = m->dev;
    v.nq = d->nq; v.ndofs = d->ndofs; v.ngeom = d->ngeom;
    v.num_cells_fe = d->num_cells;
    v.cells_per_wave = DXO_WAVE / d->nq;
    const int maxbs = G, maxD = G * (1 + G);
    int wd = v.cells_per_wave * (op_odd(d->ndofs * maxbs) + op_odd(d->ngeom * G));
    if (wd < DXO_WAVE * maxD) wd = DXO_WAVE * maxD;
    v.wave_doubles = (wd + 1) & ~1;
    v.table_doubles = (d->nq * (op_odd(d->ndofs) + op_odd(d->ndofs * G) + op_odd(d->ngeom * G)) + 1) & ~1;
    if ((size_t)(v.table_doubles + 4 * v.wave_doubles) * sizeof(double) > 64 * 1024) {
        delete m;
        return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_create: element too large for the 64 KiB LDS budget of the operand kernel");
    }
    // one blob: phi | dphi | dpsi | x | dofmap | geom_dofmap
    const size_t n_phi = (size_t)d->nq * d->ndofs, n_dphi = n_phi * G, n_dpsi = (size_t)d->nq * d->ngeom * G;
    const size_t n_x = (size_t)d->num_geom_nodes * G;
    const size_t n_dm = (size_t)d->num_cells * d->ndofs, n_gm = (size_t)d->num_cells * d->ngeom;
    const size_t bytes = (n_phi + n_dphi + n_dpsi + n_x) * sizeof(double) + (n_dm + n_gm) * sizeof(int32_t);
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipMalloc(&m->blob, bytes ? bytes : 8);
    if (e != hipSuccess) { delete m; return dxo_hip_fail(ctx, e, "dxo_mesh_create: hipMalloc"); }
    std::vector<char> host(bytes);
    char* h = host.data();
    double* hd = reinterpret_cast<double*>(h);
    std::memcpy(hd, d->phi, n_phi * sizeof(double));
    std::memcpy(hd + n_phi, d->dphi, n_dphi * sizeof(double));
    std::memcpy(hd + n_phi + n_dphi, d->dpsi, n_dpsi * sizeof(double));
    // coordinates: x_stride doubles per node in the source (DOLFINx pads to 3), gdim kept
    const int xs = d->x_stride > 0 ? d->x_stride : G;
    double* hx = hd + n_phi + n_dphi + n_dpsi;
    for (int64_t i = 0; i < d->num_geom_nodes; ++i)
        for (int j = 0; j < G; ++j) hx[i * G + j] = d->x[i * xs + j];
    int32_t* hi = reinterpret_cast<int32_t*>(hx + n_x);
    if (n_dm) std::memcpy(hi, d->dofmap, n_dm * sizeof(int32_t));
    if (n_gm) std::memcpy(hi + n_dm, d->geom_dofmap, n_gm * sizeof(int32_t));
    e = hipMemcpy(m->blob, h, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(m->blob); delete m; return dxo_hip_fail(ctx, e, "dxo_mesh_create: hipMemcpy"); }
    double* bd = static_cast<double*>(m->blob);
    v.phi = bd; v.dphi = bd + n_phi; v.dpsi = bd + n_phi + n_dphi; v.x = bd + n_phi + n_dphi + n_dpsi;
    const int32_t* bi = reinterpret_cast<const int32_t*>(v.x + n_x);
    v.dofmap = bi; v.geom_dofmap = bi + n_dm;
    if (n_dm) m->h_dofmap.assign(d->dofmap, d->dofmap + n_dm);
    m->h_cell_xyz.assign((size_t)d->num_cells * 3, 0.0f);      // vertex mean of every cell: orders the wave groups (adjoint_patch.h)
    for (int64_t c = 0; c < d->num_cells; ++c)
        for (int v2 = 0; v2 < d->ngeom; ++v2)
            for (int j = 0; j < G; ++j) m->h_cell_xyz[(size_t)c * 3 + j] += (float)(hx[(int64_t)d->geom_dofmap[c * d->ngeom + v2] * G + j] / d->ngeom);
    *out = m;
    return DXO_OK;
}

extern "C" int dxo_mesh_destroy(dxo_ctx* ctx, dxo_mesh* m) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return DXO_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (m->blob) (void)hipFree(m->blob);
    if (m->d_u) (void)hipFree(m->d_u);
    if (m->d_cells) (void)hipFree(m->d_cells);
    if (m->d_out) (void)hipFree(m->d_out);
    if (m->d_facet_tab) (void)hipFree(m->d_facet_tab);
    if (m->d_ents) (void)hipFree(m->d_ents);
    if (m->d_wq) (void)hipFree(m->d_wq);
    if (m->d_psi) (void)hipFree(m->d_psi);
    if (m->d_node_ptr) (void)hipFree(m->d_node_ptr);
    if (m->d_node_ent) (void)hipFree(m->d_node_ent);
    if (m->d_fe) (void)hipFree(m->d_fe);
    if (m->patch.blob) (void)hipFree(m->patch.blob);
    if (m->patch.dev.bpart) (void)hipFree(m->patch.dev.bpart);
    delete m;
    return DXO_OK;
}

// ---- the operand `x = ufl.SpatialCoordinate(mesh)` (the first operand of test/test_nested_ex_op.py:113-118): the coordinate element's own
// interpolation x(xi_q) = sum_v psi_v(xi_q) X_v, i.e. the VALUE operand of the "field" (geometry dofmap, coordinates, psi). The mesh keeps psi
// once it has been given; the launch is the dense vector-valued kernel on a view of the mesh whose field element is the coordinate element.
extern "C" int dxo_mesh_set_coordinate_values(dxo_ctx* ctx, dxo_mesh* m, const double* psi) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m || !psi) return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_set_coordinate_values: NULL argument");
    const size_t n = (size_t)m->dev.nq * m->dev.ngeom;
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    DXO_HIP(ctx, hipDeviceSynchronize());
    if (!m->d_psi) DXO_HIP(ctx, hipMalloc((void**)&m->d_psi, (n ? n : 1) * sizeof(double)));
    DXO_HIP(ctx, hipMemcpy(m->d_psi, psi, n * sizeof(double), hipMemcpyHostToDevice));
    return DXO_OK;
}

extern "C" int dxo_eval_coordinate(dxo_ctx* ctx, dxo_mesh* m, int mem, const int32_t* cells, int64_t n_cells, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_coordinate: mesh is NULL");
    if (!m->d_psi) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_coordinate: call dxo_mesh_set_coordinate_values first");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_eval_coordinate: bad mem");
    if (!cells) n_cells = n_cells < 0 ? m->num_cells : n_cells;
    if (n_cells < 0 || (!cells && n_cells > m->num_cells)) return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_coordinate: bad n_cells");
    if (n_cells == 0) return DXO_OK;
    if (!out) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_coordinate: NULL array");
    if ((uintptr_t)out & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_eval_coordinate: out must be 8-byte aligned");
    const int G = m->gdim;
    OperandDev dev = m->dev;
    dev.ndofs = dev.ngeom;
    dev.phi = m->d_psi;
    dev.dphi = dev.dpsi;
    dev.dofmap = dev.geom_dofmap;
    int wd = dev.cells_per_wave * 2 * op_odd(dev.ngeom * G);
    if (wd < DXO_WAVE * G * (1 + G)) wd = DXO_WAVE * G * (1 + G);
    dev.wave_doubles = (wd + 1) & ~1;
    dev.table_doubles = (dev.nq * (op_odd(dev.ngeom) + 2 * op_odd(dev.ngeom * G)) + 1) & ~1;
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const int32_t* dc = cells;
    double* dout = out;
    const size_t out_bytes = (size_t)n_cells * dev.nq * G * sizeof(double);
    if (mem == DXO_MEM_HOST) {
        if (cells) {
            for (int64_t i = 0; i < n_cells; ++i)
                if (cells[i] < 0 || cells[i] >= m->num_cells) return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_coordinate: entity outside [0, num_cells)");
            int rc = ensure(ctx, (void**)&m->d_cells, &m->cells_cap, (size_t)n_cells * sizeof(int32_t));
            if (rc != DXO_OK) return rc;
            DXO_HIP(ctx, hipMemcpyAsync(m->d_cells, cells, (size_t)n_cells * sizeof(int32_t), hipMemcpyHostToDevice, s));
            dc = m->d_cells;
        }
        int rc = ensure(ctx, (void**)&m->d_out, &m->out_cap, out_bytes);
        if (rc != DXO_OK) return rc;
        dout = m->d_out;
    }
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    if (G == 2) launch_operand_dev<2, 2, DXO_OPERAND_VALUE>(ctx, dev, dev.x, dc, n_cells, dout, s);
    else        launch_operand_dev<3, 3, DXO_OPERAND_VALUE>(ctx, dev, dev.x, dc, n_cells, dout, s);
    rc = dxo_device_end(ctx, s);
    if (rc != DXO_OK) return rc;
    if (mem == DXO_MEM_HOST) {
        DXO_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));
    }
    return DXO_OK;
}

extern "C" int dxo_eval_operand(dxo_ctx* ctx, dxo_mesh* m, int kind, int bs, int mem, const double* u,
                                const int32_t* cells, int64_t n_cells, double* out) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!m) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_operand: mesh is NULL");
    const int D = dxo_operand_value_size(m->gdim, bs, kind);
    if (D == DXO_E_OPTION) return dxo_fail(ctx, DXO_E_OPTION, "dxo_eval_operand: unknown operand kind");
    if (D < 0) return dxo_fail(ctx, DXO_E_DIM, "dxo_eval_operand: block size does not fit the operand kind / gdim");
    if (mem != DXO_MEM_HOST && mem != DXO_MEM_DEVICE) return dxo_fail(ctx, DXO_E_MEM, "dxo_eval_operand: bad mem");
    if (!cells) n_cells = n_cells < 0 ? m->num_cells : n_cells;
    if (n_cells < 0 || (!cells && n_cells > m->num_cells)) return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_operand: bad n_cells");
    if (n_cells == 0) return DXO_OK;
    if (!u || !out) return dxo_fail(ctx, DXO_E_NULL, "dxo_eval_operand: NULL array");
    if (((uintptr_t)u | (uintptr_t)out) & 7u) return dxo_fail(ctx, DXO_E_ALIGN, "dxo_eval_operand: arrays must be 8-byte aligned");
    hipStream_t s = dxo_launch_stream(ctx);
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    const double* du = u;
    const int32_t* dc = cells;
    double* dout = out;
    const size_t out_bytes = (size_t)n_cells * m->dev.nq * D * sizeof(double);
    if (mem == DXO_MEM_HOST) {
        if (cells)
            for (int64_t i = 0; i < n_cells; ++i)
                if (cells[i] < 0 || cells[i] >= m->num_cells) return dxo_fail(ctx, DXO_E_SIZE, "dxo_eval_operand: entity outside [0, num_cells)");
        const size_t ub = (size_t)m->num_field_nodes * bs * sizeof(double);
        int rc = ensure(ctx, (void**)&m->d_u, &m->u_cap, ub);
        if (rc != DXO_OK) return rc;
        DXO_HIP(ctx, hipMemcpyAsync(m->d_u, u, ub, hipMemcpyHostToDevice, s));
        du = m->d_u;
        if (cells) {
            rc = ensure(ctx, (void**)&m->d_cells, &m->cells_cap, (size_t)n_cells * sizeof(int32_t));
            if (rc != DXO_OK) return rc;
            DXO_HIP(ctx, hipMemcpyAsync(m->d_cells, cells, (size_t)n_cells * sizeof(int32_t), hipMemcpyHostToDevice, s));
            dc = m->d_cells;
        }
        rc = ensure(ctx, (void**)&m->d_out, &m->out_cap, out_bytes);
        if (rc != DXO_OK) return rc;
        dout = m->d_out;
    }
    int rc = dxo_device_begin(ctx, s);
    if (rc != DXO_OK) return rc;
    if (bs != 1 && bs != m->gdim) rc = m->gdim == 2 ? dispatch_components<2>(ctx, m, kind, bs, du, dc, n_cells, dout, s) : dispatch_components<3>(ctx, m, kind, bs, du, dc, n_cells, dout, s);
    else if (m->gdim == 2) rc = bs == 1 ? dispatch_kind<2, 1>(ctx, m, kind, du, dc, n_cells, dout, s) : dispatch_kind<2, 2>(ctx, m, kind, du, dc, n_cells, dout, s);
    else              rc = bs == 1 ? dispatch_kind<3, 1>(ctx, m, kind, du, dc, n_cells, dout, s) : dispatch_kind<3, 3>(ctx, m, kind, du, dc, n_cells, dout, s);
    if (rc != DXO_OK) return dxo_fail(ctx, rc, "dxo_eval_operand: unsupported (gdim, bs, kind)");
    rc = dxo_device_end(ctx, s);
    if (rc != DXO_OK) return rc;
    if (mem == DXO_MEM_HOST) {
        DXO_HIP(ctx, hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s));
        DXO_HIP(ctx, hipStreamSynchronize(s));
    }
    return DXO_OK;
}
