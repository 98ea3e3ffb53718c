// adjoint_patch_kernels.h — the PATCH form of the hexahedral internal force (round 5; measured slower than element vectors + node sums:
// 0.79-0.85 against 0.61 ms, profiles/r05_patch_form.txt). Not part of the product build: csrc/adjoint.hip includes this file three
// times under -DDXO_EXPERIMENTS (DXO_PATCH_PART 1: the kernel, behind operand_adjoint_c8; 2: the host side, behind ensure_transpose;
// 3: the dxo_mesh_patch_info entry point, which is not declared in include/dxo.h any more). Data structures and the LDS accumulator:
// adjoint_patch.h beside this file. In the product build ctx option adjoint_patch = 1 is refused (DXO_E_OPTION).
#if DXO_PATCH_PART == 1
// operand_adjoint_c8 in the PATCH form (adjoint_patch.h): one workgroup per patch, its waves take one wave group per iteration
// (groups of an iteration share no node), the element-vector entries are added into the patch's LDS accumulator, and only the
// nodes shared with another patch leave a partial in HBM. Same arithmetic per entry as operand_adjoint_c8; a node's entries are
// added in schedule order instead of transposed-dofmap order (the sums agree to rounding, each form is bit-reproducible).
template <int ND>
__global__ __launch_bounds__(PATCH_BLOCK) void operand_adjoint_c8_patch(OperandDev m, const double* __restrict__ wq, const double* __restrict__ S,
                                                                        PatchDev P, double* __restrict__ out, int overwrite) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    c8_fill_tables(m, lds);
    const PatchAcc<3> A{lds + C8_LDS, P.max_priv};
    const int lane = threadIdx.x & (DXO_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    double* acc = A.mine(wave);
    const C8Lane L(lds, lane);
    constexpr int cpw = 8;
    const double w_l = wq[lane & 7];
    const int c_l = lane >> 3, q_l = lane & 7;
    const int R = P.R;
    auto cells_in = [&](int32_t g) -> int {
        if (g < 0) return 0;
        const int64_t left = m.num_cells_fe - (int64_t)g * cpw;
        return left < cpw ? (int)left : cpw;
    };
    auto vertex_index = [&](int32_t g) -> int32_t { return c_l < cells_in(g) ? m.geom_dofmap[((int64_t)g * cpw + c_l) * 8 + q_l] : -1; };
    auto vertex = [&](int32_t xn, double (&xv)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) xv[j] = xn >= 0 ? m.x[(int64_t)xn * 3 + j] : 0.0;
    };
    // the wave's sequence of groups: R per patch, patches blockIdx.x, + gridDim.x, ...; the gather runs two groups ahead across patches
    auto group_at = [&](int64_t patch, int r) -> int32_t {
        return patch < P.n_patches ? P.groups[((size_t)patch * PATCH_WAVES + wave) * R + r] : -1;
    };
    int64_t patch = blockIdx.x;
    int32_t grp = group_at(patch, 0);
    int32_t xn = vertex_index(grp);
    double xv[3];
    vertex(xn, xv);
    auto advance = [&](int64_t& pp, int& rr) { if (++rr == R) { rr = 0; pp += gridDim.x; } };
    int64_t pn = patch; int rn = 0;
    advance(pn, rn);
    int32_t grp_n = group_at(pn, rn);
    xn = vertex_index(grp_n);
    for (; patch < P.n_patches; patch += gridDim.x) {
        __syncthreads();                 // tables filled (first patch) / the previous patch's merge has read the accumulators
        A.zero(wave, lane);
        for (int r = 0; r < R; ++r) {
            const int64_t c0 = (int64_t)grp * cpw;
            const int ncell = cells_in(grp);
            const bool has_point = c_l < ncell;
            const int64_t cell = c0 + c_l;
            // the lane's four nodes' entries in the wave's accumulator and its cell's colour, requested before the arithmetic
            int ln4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) ln4[t] = (has_point && L.node0 + t < ND) ? (int)P.lnode[cell * ND + L.node0 + t] : 0;
            const int col_l = has_point ? (int)P.cellcol[cell] : 0;
            const int ncol = grp >= 0 ? (int)P.grp_ncol[grp] : 0;
            dxo_f64x2 s2[3];
            {
                const dxo_f64x2* Sp = reinterpret_cast<const dxo_f64x2*>(S + (c0 * 8 + lane) * 6);
#pragma unroll
                for (int k = 0; k < 3; ++k) s2[k] = has_point ? Sp[k] : dxo_f64x2{0.0, 0.0};
            }
            double K[3][3];
            const double det = c8_geometry(L, xv, K);
            vertex(xn, xv);                              // the next group's vertex (its index has been here for an iteration)
            advance(pn, rn);
            const int32_t grp_nn = group_at(pn, rn);
            xn = vertex_index(grp_nn);
            const double s[6] = {s2[0].x, s2[0].y, s2[1].x, s2[1].y, s2[2].x, s2[2].y};
            double vh[3], gh[3][3], T[3][3];
            dual_tensor<3, 3, DXO_OPERAND_EPS_MANDEL>(s, vh, gh);
            const double scale = w_l * fabs(det);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double tt = 0.0;
#pragma unroll
                    for (int j = 0; j < 3; ++j) tt += gh[i][j] * K[k][j];
                    T[i][k] = has_point ? scale * tt : 0.0;       // lanes without a point: zero vertices, singular J
                }
            c8_scatter_all<ND>(L, T, [&](int t, const double (&o)[3]) { A.add(acc, ncol, col_l, has_point && L.node0 + t < ND, ln4[t], o); });
            grp = grp_n;
            grp_n = grp_nn;
        }
        __syncthreads();
        A.merge_flush(P, (int)patch, out, overwrite);
    }
}

#elif DXO_PATCH_PART == 2
// patch form (adjoint_patch.h): built once per mesh on first use; bpart sized for the call's block size
bool ensure_patches(dxo_ctx* ctx, dxo_mesh* m, int bs) {
    PatchSet& ps = m->patch;
    const int bs_cap = m->gdim;
    if (!ps.built) {
        ps.built = true;
        ps.cpw = m->dev.cells_per_wave;
        ps.bs_cap = bs_cap;
        PatchHost H;
        bool ok = false;
        if (m->h_cell_xyz.size() == (size_t)m->num_cells * 3)
            for (int R = DXO_PATCH_R; R >= 1 && !ok; --R)      // a wave's nodes must fit its accumulator: fewer groups per wave if not
                ok = patch_build_host(m->h_dofmap, m->h_cell_xyz, m->num_cells, m->dev.ndofs, m->num_field_nodes, ps.cpw, bs_cap, R, H);
        if (!ok) return false;
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t sz[] = {H.groups.size() * 4, H.node_off.size() * 4, H.gnode.size() * 4, H.mmap.size() * 2, H.lnode.size() * 2, H.cellcol.size(),
                             H.grp_ncol.size(), H.bnode.size() * 4, H.bptr.size() * 8, H.bent.size() * 4};
        const void* src[] = {H.groups.data(), H.node_off.data(), H.gnode.data(), H.mmap.data(), H.lnode.data(), H.cellcol.data(), H.grp_ncol.data(),
                             H.bnode.data(), H.bptr.data(), H.bent.data()};
        size_t off[10], total = 0;
        for (int k = 0; k < 10; ++k) { off[k] = total; total += al(sz[k] ? sz[k] : 1); }
        if (hipMalloc(&ps.blob, total) != hipSuccess) { (void)hipGetLastError(); ps.blob = nullptr; return false; }
        char* b = static_cast<char*>(ps.blob);
        for (int k = 0; k < 10; ++k)
            if (sz[k] && hipMemcpy(b + off[k], src[k], sz[k], hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return false; }
        PatchDev& d = ps.dev;
        d.n_patches = (int32_t)(H.node_off.size() - 1);
        d.R = H.R;
        d.max_priv = H.max_priv;
        d.max_local = H.max_local;
        d.groups = reinterpret_cast<const int32_t*>(b + off[0]);
        d.node_off = reinterpret_cast<const int32_t*>(b + off[1]);
        d.gnode = reinterpret_cast<const uint32_t*>(b + off[2]);
        d.mmap = reinterpret_cast<const uint16_t*>(b + off[3]);
        d.lnode = reinterpret_cast<const uint16_t*>(b + off[4]);
        d.cellcol = reinterpret_cast<const uint8_t*>(b + off[5]);
        d.grp_ncol = reinterpret_cast<const uint8_t*>(b + off[6]);
        d.bnode = reinterpret_cast<const int32_t*>(b + off[7]);
        d.bptr = reinterpret_cast<const int64_t*>(b + off[8]);
        d.bent = reinterpret_cast<const uint32_t*>(b + off[9]);
        d.n_bnodes = (int64_t)H.bnode.size();
        ps.n_slots = (int64_t)H.gnode.size();
        ps.shared_fraction = H.shared_fraction;
        ps.usable = true;
    }
    if (!ps.usable || bs > ps.bs_cap) return false;
    const size_t need = (size_t)ps.n_slots * bs * sizeof(double);
    if (ps.bpart_cap < need) {
        if (ps.dev.bpart) (void)hipFree(ps.dev.bpart);
        ps.dev.bpart = nullptr;
        ps.bpart_cap = 0;
        if (hipMalloc((void**)&ps.dev.bpart, need ? need : 8) != hipSuccess) { (void)hipGetLastError(); return false; }
        ps.bpart_cap = need;
    }
    return true;
}

// does this call take the patch form? (whole mesh, no entity list, not the atomics option)
bool use_patches(dxo_ctx* ctx, dxo_mesh* m, int bs, const int32_t* cells, int64_t n_cells) {
    if (!ctx->adjoint_patch || ctx->adjoint_atomics || cells || n_cells != m->num_cells) return false;
    return ensure_patches(ctx, m, bs);
}

void launch_node_sum_patch(const dxo_ctx* ctx, const dxo_mesh* m, int bs, double* out, hipStream_t s) {
    const PatchDev& P = m->patch.dev;
    int64_t blocks = (P.n_bnodes + DXO_BLOCK - 1) / DXO_BLOCK;
    if (blocks < 1) blocks = 1;
    const int ow = (int)(ctx->consumer_overwrite != 0);
    if (bs == 1) hipLaunchKernelGGL(node_sum_patch<1>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, P, out, ow);
    else if (bs == 2) hipLaunchKernelGGL(node_sum_patch<2>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, P, out, ow);
    else hipLaunchKernelGGL(node_sum_patch<3>, dim3((int)blocks), dim3(DXO_BLOCK), 0, s, P, out, ow);
}

int patch_grid(const dxo_ctx* ctx, const dxo_mesh* m, int blocks_per_cu) {
    int64_t blocks = (int64_t)ctx->compute_units * blocks_per_cu;
    if (blocks > m->patch.dev.n_patches) blocks = m->patch.dev.n_patches;
    return (int)(blocks < 1 ? 1 : blocks);
}

#elif DXO_PATCH_PART == 3
extern "C" int dxo_mesh_patch_info(dxo_ctx* ctx, dxo_mesh* mesh, int64_t info[8]) {
    if (!ctx) return DXO_E_NULL;
    DXO_LOCK(ctx);
    if (!mesh || !info) return dxo_fail(ctx, DXO_E_NULL, "dxo_mesh_patch_info: NULL argument");
    DXO_HIP(ctx, hipSetDevice(ctx->device));
    if (!ensure_patches(ctx, mesh, mesh->gdim)) return dxo_fail(ctx, DXO_E_SIZE, "dxo_mesh_patch_info: this mesh cannot use the patch form");
    const PatchSet& ps = mesh->patch;
    info[0] = ps.dev.n_patches; info[1] = ps.dev.R; info[2] = (mesh->num_cells + ps.cpw - 1) / ps.cpw; info[3] = ps.dev.max_local;
    info[4] = ps.n_slots; info[5] = ps.dev.n_bnodes; info[6] = (int64_t)(ps.shared_fraction * 1e6); info[7] = PATCH_WAVES;
    info[1] = info[1] | ((int64_t)ps.dev.max_priv << 32);
    return DXO_OK;
}

#endif
