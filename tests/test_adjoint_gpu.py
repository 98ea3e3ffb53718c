"""The consumer side on the device: dxo_operand_adjoint (assembled vector of inner(S, operand(v)) dx)
and dxo_tangent_apply (matrix-free K v). No reference code exists for these on the GPU box (DOLFINx assembly), so the
pins are mathematical identities that do not depend on any oracle — adjointness <B u, S>_w = <u, B^T S>, the patch test,
symmetry of v -> K v — plus agreement with the NumPy oracle."""
import numpy as np
import pytest

from tools.synthetic import structured_mesh
from oracle.operand_oracle import (DEFGRAD, EPS_MANDEL, GRAD, VALUE, VALUE_GRAD, _geometry, eval_operand, operand_adjoint,
                                   tangent_apply)

pytestmark = pytest.mark.gpu
CELLS = {"triangle": (6, 5), "quadrilateral": (4, 4), "tetrahedron": (2, 3, 2), "hexahedron": (3, 2, 3)}
KIND_ID = {"value": VALUE, "grad": GRAD, "eps": EPS_MANDEL, "F": DEFGRAD, "value_grad": VALUE_GRAD, "div": 8}


def device_adjoint(ctx, dm, kind, bs, S, n_nodes, cells=None):
    import torch

    St = torch.from_numpy(np.ascontiguousarray(S).reshape(-1)).cuda()
    out = torch.zeros(n_nodes * bs, dtype=torch.float64, device="cuda")
    ct = None if cells is None else torch.from_numpy(np.ascontiguousarray(cells, dtype=np.int32)).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dm.adjoint(kind, bs, St.data_ptr(), out.data_ptr(), n_cells=None if cells is None else len(cells),
               cells_ptr=None if ct is None else ct.data_ptr())
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_adjoint_identity_patch_test_and_oracle(ctx, cell, degree):
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh(cell, CELLS[cell], degree, distort=0.2, seed=6)
    G, nn = m.gdim, m.node_x.shape[0]
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    rng = np.random.Generator(np.random.PCG64(3))
    _, det = _geometry(m.dofmap, m.geom_dofmap, m.x, m.dphi, m.dpsi, np.arange(m.num_cells))
    wdet = m.weights[None, :] * np.abs(det)
    try:
        for bs in (1, G):
            for kind in ("value", "grad", "value_grad") + (("eps", "F", "div") if bs == G else ()):
                u = rng.normal(size=nn * bs)
                e = dm.evaluate(kind, bs, u)                               # B u on the device
                if kind == "F":
                    e = e - np.eye(G).reshape(-1)                          # the adjoint is that of the linear part
                S = rng.normal(size=e.shape)
                f = device_adjoint(ctx, dm, kind, bs, S, nn)
                lhs, rhs = np.sum(wdet[:, :, None] * e * S), u @ f          # <B u, S>_w  vs  <u, B^T S>
                assert abs(lhs - rhs) <= 1e-12 * max(abs(lhs), np.abs(wdet).sum()), (kind, bs, lhs, rhs)
                ref = operand_adjoint(KIND_ID[kind], bs, S, m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn)
                assert np.abs(f - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30), (kind, bs)
        # the two forms of the scatter: element vectors + node sums (default, no atomics: bit-reproducible) and fp64
        # atomics into the dof vector (entity subsets always use it)
        S = rng.normal(size=(m.num_cells, m.nq, dm.value_size("eps", G)))
        two_a, two_b = device_adjoint(ctx, dm, "eps", G, S, nn), device_adjoint(ctx, dm, "eps", G, S, nn)
        assert np.array_equal(two_a, two_b)
        ctx.set_option("adjoint_atomics", 1)
        try:
            atom = device_adjoint(ctx, dm, "eps", G, S, nn)
        finally:
            ctx.set_option("adjoint_atomics", 0)
        assert np.abs(atom - two_a).max() <= 1e-13 * np.abs(two_a).max()
        # patch test: a constant stress field does no work on interior nodes
        Sc = np.broadcast_to(rng.normal(size=dm.value_size("eps", G)), (m.num_cells, m.nq, dm.value_size("eps", G)))
        fc = device_adjoint(ctx, dm, "eps", G, Sc, nn).reshape(-1, G)
        interior = np.all((m.node_x > 1e-9) & (m.node_x < 1 - 1e-9), axis=1)
        if interior.any():
            assert np.abs(fc[interior]).max() <= 1e-13 * np.abs(fc).max()
        # an entity subset: S is indexed by position in the entity list, like Expression.eval's output
        cells = np.array([m.num_cells - 1, 0, 2], dtype=np.int32)
        Ss = rng.normal(size=(3, m.nq, G))
        fs = device_adjoint(ctx, dm, "grad", 1, Ss, nn, cells)
        ref = operand_adjoint(GRAD, 1, Ss, m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn, cells)
        assert np.abs(fs - ref).max() <= 1e-12 * np.abs(ref).max()
    finally:
        dm.close()


@pytest.mark.parametrize("cell,n", [("triangle", (9, 8)), ("hexahedron", (4, 3, 3)), ("tetrahedron", (2, 2, 3))])
def test_matrix_free_tangent(ctx, oracle, cell, n):
    """K v from dxo_tangent_apply with the tangents the von Mises kernel wrote: equals the oracle's B^T C B v, is
    symmetric (w.Kv = v.Kw), and with C = C_elas a linear displacement field loads only the boundary."""
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE, DeviceMesh, VmParams

    m = structured_mesh(cell, n, 2, distort=0.15, seed=8)
    G, nn = m.gdim, m.node_x.shape[0]
    d = 4 if G == 2 else 6
    npts = m.num_cells * m.nq
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    rng = np.random.Generator(np.random.PCG64(5))
    E = 70e3
    prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
    try:
        u = rng.normal(size=nn * G)
        u *= 1.5e-3 / eval_operand(EPS_MANDEL, G, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi).std()
        sig_n, p = rng.normal(0.0, 50.0, npts * d), np.abs(rng.normal(0.0, 1e-3, npts))
        t = {k: torch.from_numpy(v).cuda() for k, v in (("u", u), ("sn", sig_n), ("p", p))}
        C = torch.empty(npts * d * d, dtype=torch.float64, device="cuda")
        s = torch.empty(npts * d, dtype=torch.float64, device="cuda")
        dp = torch.empty(npts, dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dm.von_mises(prm, t["u"].data_ptr(), t["sn"].data_ptr(), t["p"].data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr(), mem=MEM_DEVICE)
        torch.cuda.synchronize()
        assert 0.1 < float((dp > 0).double().mean()) < 0.99
        Ch = C.cpu().numpy()

        def K_times(vec):
            vt = torch.from_numpy(np.ascontiguousarray(vec)).cuda()
            out = torch.zeros(nn * G, dtype=torch.float64, device="cuda")
            dm.tangent_apply(C.data_ptr(), vt.data_ptr(), out.data_ptr())
            torch.cuda.synchronize()
            return out.cpu().numpy()

        v, w = rng.normal(size=nn * G), rng.normal(size=nn * G)
        Kv, Kw = K_times(v), K_times(w)
        ref = tangent_apply(Ch, v, m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn)
        assert np.abs(Kv - ref).max() <= 1e-12 * np.abs(ref).max()
        # the specialised forms (lane = cell on P2 triangles, register scatter on hexahedra keep running; the internal force's
        # specialised kernels are switched off by the same option) against the generic wave-group kernels
        ctx.set_option("adjoint_cell", 0)
        try:
            Kv_generic = K_times(v)
        finally:
            ctx.set_option("adjoint_cell", 1)
        assert np.abs(Kv - Kv_generic).max() <= 1e-13 * np.abs(ref).max()
        assert np.array_equal(K_times(v), Kv)                               # bit-reproducible
        # the same operator WITHOUT the tangent array: K v and diag(K) from the returned (sigma, dp) (dxo_tangent_apply_vm /
        # dxo_tangent_diagonal_vm), and the fused operator run with C_tang = NULL leaves the same (sigma, dp)
        vt = torch.from_numpy(v).cuda()
        out_vm = torch.zeros(nn * G, dtype=torch.float64, device="cuda")
        dm.tangent_apply_vm(prm, s.data_ptr(), dp.data_ptr(), vt.data_ptr(), out_vm.data_ptr())
        torch.cuda.synchronize()
        assert np.abs(out_vm.cpu().numpy() - Kv).max() <= 1e-13 * np.abs(ref).max()
        dg, dg_vm = (torch.zeros(nn * G, dtype=torch.float64, device="cuda") for _ in range(2))
        dm.tangent_diagonal(C.data_ptr(), dg.data_ptr())
        dm.tangent_diagonal_vm(prm, s.data_ptr(), dp.data_ptr(), dg_vm.data_ptr())
        torch.cuda.synchronize()
        assert float((dg - dg_vm).abs().max()) <= 1e-13 * float(dg.abs().max()) and float(dg.min()) > 0
        s2, dp2 = torch.empty_like(s), torch.empty_like(dp)
        dm.von_mises(prm, t["u"].data_ptr(), t["sn"].data_ptr(), t["p"].data_ptr(), None, s2.data_ptr(), dp2.data_ptr(), mem=MEM_DEVICE)
        torch.cuda.synchronize()
        assert torch.equal(s2, s) and torch.equal(dp2, dp)
        assert abs(w @ Kv - v @ Kw) <= 1e-11 * abs(w @ Kv)                  # the consistent tangent is symmetric
        assert v @ Kv > 0                                                   # and positive here (hardening material)
        # internal force from the stresses of the same launch, against the oracle
        f = torch.zeros(nn * G, dtype=torch.float64, device="cuda")
        dm.adjoint("eps", G, s.data_ptr(), f.data_ptr())
        torch.cuda.synchronize()
        fref = operand_adjoint(EPS_MANDEL, G, s.cpu().numpy(), m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn)
        assert np.abs(f.cpu().numpy() - fref).max() <= 1e-12 * np.abs(fref).max()
        # elastic patch test: C = C_elas everywhere, v linear in x -> constant stress -> interior rows of K v vanish
        lm, mu = E * 0.3 / (1.3 * 0.4), E / 2.6
        one = np.zeros(d)
        one[:3] = 1.0
        C_el = lm * np.outer(one, one) + 2 * mu * np.eye(d)
        C.copy_(torch.from_numpy(np.tile(C_el.reshape(-1), npts)).cuda())
        A = rng.normal(size=(G, G))
        v_lin = (m.node_x @ A.T).reshape(-1)
        Kl = K_times(v_lin).reshape(-1, G)
        interior = np.all((m.node_x > 1e-9) & (m.node_x < 1 - 1e-9), axis=1)
        assert np.abs(Kl[interior]).max() <= 1e-12 * np.abs(Kl).max()
    finally:
        dm.close()


def test_weights_are_required(ctx):
    import torch

    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh("triangle", (2, 2), 1)
    dm = DeviceMesh(gdim=2, phi=m.phi, dphi=m.dphi, dpsi=m.dpsi, dofmap=m.dofmap, geom_dofmap=m.geom_dofmap, x=m.x,
                    num_field_nodes=m.node_x.shape[0], ctx=ctx)
    try:
        S = torch.zeros(m.num_cells * m.nq * 2, dtype=torch.float64, device="cuda")
        out = torch.zeros(m.node_x.shape[0], dtype=torch.float64, device="cuda")
        with pytest.raises(ValueError):
            dm.adjoint("grad", 1, S.data_ptr(), out.data_ptr())
        with pytest.raises(ValueError):
            dm.set_weights(np.ones(m.nq + 1))
        dm.set_weights(m.weights)
        dm.adjoint("grad", 1, S.data_ptr(), out.data_ptr())
    finally:
        dm.close()


@pytest.mark.parametrize("tangent_array,graph", [(False, False), (True, False), (False, True)])
def test_device_newton_krylov_converges_quadratically(tangent_array, graph):
    """examples/device_newton_krylov.py: load stepping with the fused constitutive kernel, the internal force and the
    matrix-free tangent, all on the device. Newton only converges quadratically if the tangent IS the derivative of the
    stress that the residual is built from — the kernel-level version of the reference's Taylor test
    (demo_plasticity_mohr_coulomb.py:1149-1235). Both forms of the tangent: acting from the returned (sigma, dp) with no
    tangent array at all (the default), and read from the C_tang block the operator wrote; graph: the CG iteration (library call +
    torch vector updates) captured once in a HIP graph and replayed — the device entry points are capture-safe after their first call."""
    import importlib.util
    import pathlib

    path = pathlib.Path(__file__).resolve().parents[1] / "examples" / "device_newton_krylov.py"
    spec = importlib.util.spec_from_file_location("nk_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(20, verbose=False, tangent_array=tangent_array, graph=graph)
    assert rep["steps"][0]["newton_residuals"][-1] <= 1e-8 * rep["steps"][0]["newton_residuals"][0]
    assert len(rep["steps"][0]["newton_residuals"]) == 2                       # elastic step: one linear solve
    last = rep["steps"][-1]
    r = last["newton_residuals"]
    assert last["plastic_fraction"] > 0.2 and r[-1] <= 1e-8 * r[0] and len(r) <= 12
    # quadratic tail (before the round-off / CG-tolerance floor of the very last iterate): rho_{k+1} <= c rho_k^2
    rho = [v / r[0] for v in r]
    assert rho[-2] <= 50.0 * rho[-3] ** 2


@pytest.mark.parametrize("cell,n", [("triangle", (4, 3)), ("hexahedron", (2, 1, 2)), ("quadrilateral", (3, 2))])
def test_matrix_free_diagonal(ctx, cell, n):
    """diag(K) from dxo_tangent_diagonal equals e_k . K e_k with K v from the oracle, for every dof of a small mesh, with
    non-constant symmetric positive tangents."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh(cell, n, 2, distort=0.2, seed=4)
    G, nn = m.gdim, m.node_x.shape[0]
    d = 4 if G == 2 else 6
    npts = m.num_cells * m.nq
    rng = np.random.Generator(np.random.PCG64(9))
    A = rng.normal(size=(npts, d, d))
    Cn = np.einsum("nij,nkj->nik", A, A) + 3.0 * np.eye(d)              # SPD, different at every point
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        Ct = torch.from_numpy(Cn.reshape(-1)).cuda()
        out = torch.zeros(nn * G, dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dm.tangent_diagonal(Ct.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = np.empty(nn * G)
        for k in range(nn * G):
            e = np.zeros(nn * G)
            e[k] = 1.0
            want[k] = tangent_apply(Cn, e, m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn)[k]
        assert np.all(got > 0)
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    finally:
        dm.close()


@pytest.mark.parametrize("cell,n,degree", [("hexahedron", (3, 2, 2), 2), ("hexahedron", (3, 2, 2), 1), ("tetrahedron", (2, 2, 2), 2),
                                           ("triangle", (5, 4), 2), ("quadrilateral", (4, 3), 2)])
def test_lane_per_cell_kernel_equals_wave_group_kernel(ctx, cell, n, degree):
    """The standard elements take the lane = cell kernel for the virtual work of eps (adjoint_cell.h); switching it off
    (option adjoint_cell = 0) must give the same vector to rounding."""
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh(cell, n, degree, distort=0.2, seed=12)
    G, nn = m.gdim, m.node_x.shape[0]
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    rng = np.random.Generator(np.random.PCG64(1))
    S = rng.normal(size=(m.num_cells, m.nq, dm.value_size("eps", G)))
    try:
        a = device_adjoint(ctx, dm, "eps", G, S, nn)
        ctx.set_option("adjoint_cell", 0)
        try:
            b = device_adjoint(ctx, dm, "eps", G, S, nn)
        finally:
            ctx.set_option("adjoint_cell", 1)
        assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
        ref = operand_adjoint(EPS_MANDEL, G, S, m.weights, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, nn)
        assert np.abs(a - ref).max() <= 1e-12 * np.abs(ref).max()
    finally:
        dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n, overwrite", [((5, 4, 3), 0), ((12, 9, 7), 1), ((20, 20, 20), 1)])
def test_patch_form_of_the_internal_force_equals_the_two_pass_form(ctx, n, overwrite, experiments_build):
    """Experiments build only (scripts/exp/adjoint_patch.h, -DDXO_EXPERIMENTS: measured slower, profiles/r05_patch_form.txt); the product
    library refuses the option, which is what this test checks there. Option adjoint_patch = 1: the element-vector entries of Q2 hexahedra are added in
    LDS patch by patch and only patch-border partials go through HBM. Same sums in another (fixed) order: equal to the two-pass form
    to rounding, identical bits run to run, untouched entries handled like the two-pass form does (accumulate / overwrite)."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh
    from tools.synthetic import structured_mesh

    if not experiments_build:
        with pytest.raises(ValueError, match="adjoint_patch"):
            ctx.set_option("adjoint_patch", 1)
        assert ctx.get_option("adjoint_patch") == 0
        return
    m = structured_mesh("hexahedron", n, 2, distort=0.2, seed=3)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
    S = torch.randn(npts * 6, generator=g, device=dev, dtype=torch.float64)
    saved = (ctx.get_option("adjoint_patch"), ctx.get_option("consumer_overwrite"))
    try:
        ctx.set_option("consumer_overwrite", overwrite)
        outs = []
        for mode in (0, 1, 1):
            ctx.set_option("adjoint_patch", mode)
            out = torch.full((nn * 3,), 2.5, dtype=torch.float64, device=dev)
            dm.adjoint("eps", 3, S.data_ptr(), out.data_ptr())
            torch.cuda.synchronize()
            outs.append(out)
        info = dm.patch_info()
    finally:
        ctx.set_option("adjoint_patch", saved[0])
        ctx.set_option("consumer_overwrite", saved[1])
    scale = float(outs[0].abs().max())
    assert float((outs[1] - outs[0]).abs().max()) <= 1e-13 * scale
    assert torch.equal(outs[1], outs[2])
    assert info["patches"] >= 1 and info["wave_groups"] == -(-m.num_cells // 8) and 0 < info["schedule_fill"] <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n, degree, atomics", [((5, 4, 3), 2, 0), ((12, 9, 7), 2, 0), ((7, 5, 3), 1, 0), ((6, 5, 5), 2, 1), ((20, 20, 21), 2, 0)])
def test_matrix_pipe_scatter_equals_the_dpp_scatter_on_hexahedra(ctx, n, degree, atomics):
    """Option adjoint_mfma (default 1): on hexahedra with the 2x2x2 rule the element vectors of a wave's 8 cells are formed by 24
    v_mfma_f64_16x16x4_f64 (csrc/adjoint.hip c8m_contract) instead of the DPP reduce-scatter (0). The same sums in another fixed order:
    equal to rounding, identical bits run to run; cell counts that are not a multiple of 8, Q1 and Q2, the atomics form.
    This test compares HIP with HIP (two forms of the scatter). What pins the DEFAULT (matrix-pipe) form to the NumPy oracle are the
    tests above, which run with the library's defaults: test_adjoint_identity_patch_test_and_oracle (adjointness, patch test, oracle on
    all four cell types), test_matrix_free_tangent / _diagonal (oracle) and the Newton-Krylov convergence test."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh, VmParams
    from tools.synthetic import structured_mesh

    m = structured_mesh("hexahedron", n, degree, distort=0.2, seed=4)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
    S = torch.randn(npts * 6, generator=g, device=dev, dtype=torch.float64)
    CT = torch.randn(npts, 36, generator=g, device=dev, dtype=torch.float64)
    v = torch.randn(nn * 3, generator=g, device=dev, dtype=torch.float64)
    dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
    prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
    calls = {"force": lambda o: dm.adjoint("eps", 3, S.data_ptr(), o.data_ptr()),
             "apply": lambda o: dm.tangent_apply(CT.data_ptr(), v.data_ptr(), o.data_ptr()),
             "apply_vm": lambda o: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), o.data_ptr())}
    # the diagonal's 48-row product (two passes of c8m_contract against the tables of c8m_fill_A2), state-based form; Q1: the same kernels with 8 nodes
    calls["diag_vm"] = lambda o: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), o.data_ptr())
    if degree == 2:
        calls["diag"] = lambda o: dm.tangent_diagonal(CT.data_ptr(), o.data_ptr())        # 66 KB of LDS per workgroup: the raised launch limit
    assert ctx.get_option("adjoint_mfma") == 1
    saved = ctx.get_option("adjoint_atomics")
    try:
        ctx.set_option("adjoint_atomics", atomics)
        for name, f in calls.items():
            outs = []
            for mode in (0, 1, 1):
                ctx.set_option("adjoint_mfma", mode)
                out = torch.full((nn * 3,), 0.5, dtype=torch.float64, device=dev)
                f(out)
                torch.cuda.synchronize()
                outs.append(out)
            scale = float(outs[0].abs().max())
            assert float((outs[1] - outs[0]).abs().max()) <= 1e-13 * scale, name
            if not atomics:
                assert torch.equal(outs[1], outs[2]), name
    finally:
        ctx.set_option("adjoint_mfma", 1)
        ctx.set_option("adjoint_atomics", saved)
        dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cell, n, atomics", [("triangle", (7, 5), 0), ("triangle", (40, 33), 0), ("tetrahedron", (3, 2, 2), 0), ("tetrahedron", (9, 7, 6), 0),
                                              ("tetrahedron", (4, 3, 3), 1), ("triangle", (11, 9), 1)])
def test_matrix_pipe_scatter_on_p2_triangles_and_tetrahedra(ctx, cell, n, atomics):
    """scatter_mfma.h: on P2 triangles (3-point rule) and P2 tetrahedra (4-point rule) the state-based tangent action and diagonal form the element vectors
    of a wave's 21 / 16 cells as 6 / 9 v_mfma_f64_16x16x4_f64 (option adjoint_mfma = 1, the default) instead of the lane = (cell, node) loop
    over tensors parked in LDS (0): equal to rounding, identical bits run to run, ragged last groups, the atomics form."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh, VmParams
    from tools.synthetic import structured_mesh

    m = structured_mesh(cell, n, 2, distort=0.2, seed=7)
    G, d = m.gdim, 4 if m.gdim == 2 else 6
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    dev = torch.device("cuda", ctx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
    S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
    v = torch.randn(nn * G, generator=g, device=dev, dtype=torch.float64)
    dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
    prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
    saved = ctx.get_option("adjoint_atomics")
    try:
        ctx.set_option("adjoint_atomics", atomics)
        calls = {"apply_vm": lambda o: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), o.data_ptr()),
                 "diag_vm": lambda o: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), o.data_ptr())}      # rows (q, pair) in passes of three pairs
        for name, f in calls.items():
            outs = []
            for mode in (0, 1, 1):
                ctx.set_option("adjoint_mfma", mode)
                out = torch.full((nn * G,), 0.25, dtype=torch.float64, device=dev)
                f(out)
                torch.cuda.synchronize()
                outs.append(out)
            scale = float(outs[0].abs().max())
            assert float((outs[1] - outs[0]).abs().max()) <= 1e-13 * scale, name
            if not atomics:
                assert torch.equal(outs[1], outs[2]), name
    finally:
        ctx.set_option("adjoint_mfma", 1)
        ctx.set_option("adjoint_atomics", saved)
        dm.close()
