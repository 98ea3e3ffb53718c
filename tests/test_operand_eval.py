"""Operand evaluation at quadrature points: synthetic tables, the NumPy oracle against known
answers (polynomial fields are represented exactly, so their analytic gradients are the expected output), and the
HIP kernel (through the C ABI) against the oracle."""
import numpy as np
import pytest

from tools.synthetic import LagrangeElement, quadrature_degree2, structured_mesh
from oracle.operand_oracle import DEFGRAD, EPS_MANDEL, GRAD, VALUE, eval_operand

CELLS = {"triangle": (5, 4), "quadrilateral": (4, 3), "tetrahedron": (2, 3, 2), "hexahedron": (3, 2, 2)}
KIND_ID = {"value": VALUE, "grad": GRAD, "eps": EPS_MANDEL, "F": DEFGRAD, "value_grad": 4, "C": 5, "I1": 6, "detF": 7, "div": 8}


def poly_field(gdim, bs, degree, seed):
    """u_i(x) = a_i + b_i.x + x^T Q_i x (Q = 0 for degree 1) and its gradient."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a, b = rng.normal(size=bs), rng.normal(size=(bs, gdim))
    Q = rng.normal(size=(bs, gdim, gdim)) * (degree >= 2)
    Q = 0.5 * (Q + Q.transpose(0, 2, 1))

    def u(x):
        return a + x @ b.T + np.einsum("...j,ijk,...k->...i", x, Q, x)

    def grad(x):
        return b + 2.0 * np.einsum("ijk,...k->...ij", Q, x)

    return u, grad


def expected(kind, g, val):
    gdim = g.shape[-1]
    r = np.sqrt(2.0) * 0.5
    if kind == "value":
        return val
    if kind == "grad":
        return g.reshape(*g.shape[:2], -1)
    if kind == "F":
        return (g + np.eye(gdim)).reshape(*g.shape[:2], -1)
    if kind == "div":
        return np.einsum("...ii->...", g)[..., None]
    if kind in ("C", "I1", "detF"):      # the reference's own operand test: F = Identity(d) + grad(u); C = F.T * F; J = det(F); I1 = tr(C)
        F = g + np.eye(gdim)             # (test/test_operands_evaluation.py:32-36)
        if kind == "C":
            return np.einsum("...ki,...kj->...ij", F, F).reshape(*g.shape[:2], -1)
        if kind == "I1":
            return np.einsum("...ij,...ij->...", F, F)[..., None]
        return np.linalg.det(F)[..., None]
    if gdim == 2:
        return np.stack([g[..., 0, 0], g[..., 1, 1], 0 * g[..., 0, 0], r * (g[..., 0, 1] + g[..., 1, 0])], axis=-1)
    return np.stack([g[..., 0, 0], g[..., 1, 1], g[..., 2, 2], r * (g[..., 0, 1] + g[..., 1, 0]),
                     r * (g[..., 0, 2] + g[..., 2, 0]), r * (g[..., 1, 2] + g[..., 2, 1])], axis=-1)


@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_lagrange_tables(cell, degree):
    fe = LagrangeElement(cell, degree)
    phi, dphi = fe.tabulate(fe.nodes)
    assert np.allclose(phi, np.eye(phi.shape[0]), atol=1e-12)            # nodal basis
    pts, w = quadrature_degree2(cell)
    phi, dphi = fe.tabulate(pts)
    assert np.allclose(phi.sum(axis=1), 1.0, atol=1e-13) and np.allclose(dphi.sum(axis=1), 0.0, atol=1e-12)
    vol = {"triangle": 0.5, "quadrilateral": 1.0, "tetrahedron": 1 / 6, "hexahedron": 1.0}[cell]
    assert np.isclose(w.sum(), vol)
    # the rule integrates x_0^2 exactly (degree 2)
    exact = {"triangle": 1 / 12, "quadrilateral": 1 / 3, "tetrahedron": 1 / 60, "hexahedron": 1 / 3}[cell]
    assert np.isclose((w * pts[:, 0] ** 2).sum(), exact)


@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_oracle_known_answers_on_distorted_meshes(cell, degree):
    m = structured_mesh(cell, CELLS[cell], degree, distort=0.25, seed=3)
    xq = m.physical_points()
    for bs in (1, m.gdim):
        u, grad = poly_field(m.gdim, bs, degree, seed=bs)
        uvec = u(m.node_x).reshape(-1)                                     # nodal interpolation is exact
        for kind in ("value", "grad") + (("eps", "F", "C", "I1", "detF", "div") if bs == m.gdim else ()):
            got = eval_operand(KIND_ID[kind], bs, uvec, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
            want = expected(kind, grad(xq), u(xq))
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= 5e-12 * max(1.0, np.abs(want).max()), (cell, degree, bs, kind)
    cells = np.array([m.num_cells - 1, 0, 2], dtype=np.int32)
    sub = eval_operand(GRAD, 1, u(m.node_x)[:, 0], m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, cells)
    full = eval_operand(GRAD, 1, u(m.node_x)[:, 0], m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
    assert np.array_equal(sub, full[cells])


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(params=[1, 0], ids=["lane=cell", "wave-group"])
def strain_kernels(request, ctx):
    """Both kernel families of dxo_eval_operand's strain evaluation: operand_cell.h (lane = cell, default on the 2-D
    standard elements) and the wave-group kernel of operand_core.h (every other element / kind, ctx option operand_cell = 0)."""
    old = ctx.get_option("operand_cell")
    ctx.set_option("operand_cell", request.param)
    yield request.param
    ctx.set_option("operand_cell", old)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_hip_matches_oracle_and_known_answers(ctx, cell, degree, strain_kernels):
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh(cell, CELLS[cell], degree, distort=0.25, seed=4)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    xq = m.physical_points()
    rng = np.random.Generator(np.random.PCG64(8))
    try:
        for bs in (1, m.gdim):
            u, grad = poly_field(m.gdim, bs, degree, seed=10 + bs)
            uvec = u(m.node_x).reshape(-1)
            rough = rng.normal(size=uvec.size)                             # not a polynomial: oracle comparison only
            for kind in ("value", "grad") + (("eps", "F", "C", "I1", "detF", "div") if bs == m.gdim else ()):
                got = dm.evaluate(kind, bs, uvec)
                want = expected(kind, grad(xq), u(xq))
                assert got.shape == want.shape
                assert np.abs(got - want).max() <= 5e-12 * max(1.0, np.abs(want).max()), (cell, degree, bs, kind)
                ref = eval_operand(KIND_ID[kind], bs, rough, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
                got = dm.evaluate(kind, bs, rough)
                assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (cell, degree, bs, kind)
        # entity subsets (evaluate_operands' `entities`, external_operator.py:365-371, :402), incl. repeats and 1 cell
        for cells in (np.array([m.num_cells - 1, 0, 2, 2], dtype=np.int32), np.array([1], dtype=np.int32),
                      np.arange(m.num_cells - 1, -1, -1, dtype=np.int32)):
            got = dm.evaluate("grad", 1, rough[: m.node_x.shape[0]], cells)
            ref = eval_operand(GRAD, 1, rough[: m.node_x.shape[0]], m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, cells)
            assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
        assert dm.evaluate("value", 1, rough[: m.node_x.shape[0]], np.empty(0, dtype=np.int32)).shape == (0, m.nq, 1)
    finally:
        dm.close()


@pytest.mark.gpu
def test_argument_checks(ctx):
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh("triangle", (2, 2), 2)
    bad = m.dofmap.copy()
    bad[1, 2] = m.node_x.shape[0]
    with pytest.raises(ValueError):
        DeviceMesh(gdim=2, phi=m.phi, dphi=m.dphi, dpsi=m.dpsi, dofmap=bad, geom_dofmap=m.geom_dofmap, x=m.x,
                   num_field_nodes=m.node_x.shape[0], ctx=ctx)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        with pytest.raises(ValueError):
            dm.evaluate("eps", 2, np.zeros(5))                       # wrong field size
        with pytest.raises(ValueError):
            dm.evaluate("eps", 1, np.zeros(m.node_x.shape[0]))       # eps of a scalar field
        with pytest.raises(ValueError):
            dm.evaluate("grad", 1, np.zeros(m.node_x.shape[0]), np.array([m.num_cells], dtype=np.int32))
    finally:
        dm.close()


@pytest.mark.gpu
def test_padded_dolfinx_style_coordinates(ctx):
    """DOLFINx stores geometry.x with 3 columns also in 2-D: x_stride = 3."""
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh("triangle", (3, 3), 2, distort=0.2, seed=1)
    x3 = np.zeros((m.x.shape[0], 3))
    x3[:, :2] = m.x
    dm = DeviceMesh(gdim=2, phi=m.phi, dphi=m.dphi, dpsi=m.dpsi, dofmap=m.dofmap, geom_dofmap=m.geom_dofmap, x=x3,
                    num_field_nodes=m.node_x.shape[0], ctx=ctx)
    try:
        u, grad = poly_field(2, 2, 2, seed=0)
        got = dm.evaluate("eps", 2, u(m.node_x).reshape(-1))
        assert np.abs(got - expected("eps", grad(m.physical_points()), None)).max() <= 5e-12
    finally:
        dm.close()


@pytest.mark.gpu
def test_operand_feeds_the_von_mises_operator_like_the_reference_loop(ctx, oracle):
    """The demo's sequence (demo_plasticity_von_mises.py:445-456) with the operand evaluated on the device:
    evaluate_operands -> evaluate_external_operators, against operand oracle -> von Mises oracle."""
    from dolfinx_external_operator_amd import (DeviceMesh, QuadratureExternalOperator, evaluate_external_operators,
                                               evaluate_operands, make_von_mises)

    m = structured_mesh("triangle", (12, 9), 2, distort=0.2, seed=5)
    rng = np.random.Generator(np.random.PCG64(2))
    Du = rng.normal(0.0, 1e-4, size=m.node_x.shape[0] * 2)
    n = m.num_cells * m.nq
    sigma_n, p = rng.normal(0.0, 50.0, (n, 4)), np.abs(rng.normal(0.0, 1e-3, n))
    sigma_n[:, 2] = 0.3 * (sigma_n[:, 0] + sigma_n[:, 1])
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        holder = {"Du": Du}
        deps = dm.operand("eps", lambda: holder["Du"])
        op = QuadratureExternalOperator(deps, num_cells=m.num_cells, num_points=m.nq, value_shape=(4, 4),
                                        external_function=make_von_mises(sigma_n, p, ctx=ctx), derivatives=(1,))
        ev = evaluate_operands([op])
        assert ev[deps].shape == (m.num_cells, m.nq, 4) and deps.eval_count == 1
        ((C_tang, sigma, dp),) = evaluate_external_operators([op], ev)
        e_ref = eval_operand(EPS_MANDEL, 2, Du, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
        C_o, s_o, dp_o = oracle.von_mises(e_ref.reshape(-1, 4), sigma_n, p)
        assert 0.05 < (dp_o > 0).mean() < 0.95
        assert np.abs(sigma.reshape(-1, 4) - s_o).max() <= 1e-12 * np.abs(s_o).max()
        assert np.abs(C_tang.reshape(-1, 4, 4) - C_o).max() <= 1e-12 * np.abs(C_o).max()
        assert np.array_equal(op.ref_coefficient.x.array, C_tang)
        holder["Du"] = 2.0 * Du                                     # the field changed: the operand must follow
        assert np.abs(deps.eval(None) - 2.0 * e_ref).max() <= 1e-13 * np.abs(e_ref).max() * 2
    finally:
        dm.close()


def _vm_state(n, d, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    sigma_n, p = rng.normal(0.0, 50.0, (n, d)), np.abs(rng.normal(0.0, 1e-3, n))
    return sigma_n, p


@pytest.mark.gpu
@pytest.mark.parametrize("cell,n,chunk", [("triangle", (7, 5), 0), ("triangle", (40, 33), 640), ("quadrilateral", (9, 4), 0),
                                          ("tetrahedron", (3, 2, 2), 0), ("hexahedron", (5, 3, 3), 0),
                                          ("hexahedron", (12, 10, 9), 2048)])
def test_fused_operand_plus_von_mises(ctx, oracle, cell, n, chunk):
    """dxo_von_mises_field == operand oracle -> von Mises oracle, host arrays (optionally through several pipeline
    chunks whose borders fall inside wave groups) and device pointers."""
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE, DeviceMesh, VmParams

    E = 70e3
    prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
    m = structured_mesh(cell, n, 2, distort=0.2, seed=9)
    d = 4 if m.gdim == 2 else 6
    npts = m.num_cells * m.nq
    rng = np.random.Generator(np.random.PCG64(4))
    u = rng.normal(size=m.node_x.shape[0] * m.gdim)
    u *= 1.5e-3 / eval_operand(EPS_MANDEL, m.gdim, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi).std()
    sigma_n, p = _vm_state(npts, d, seed=1)
    e_ref = eval_operand(EPS_MANDEL, m.gdim, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
    C_o, s_o, dp_o = oracle.von_mises(e_ref.reshape(-1, d), sigma_n, p)
    assert 0.05 < (dp_o > 0).mean() < 0.98
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    old_chunk = ctx.get_option("host_chunk_points")
    try:
        if chunk:
            ctx.set_option("host_chunk_points", chunk)
        C, s, dp = np.full(npts * d * d + 4, -7.0), np.full(npts * d + 4, -7.0), np.full(npts + 4, -7.0)
        dm.von_mises(prm, u, sigma_n, p, C, s, dp)
        assert np.all(C[npts * d * d:] == -7.0) and np.all(s[npts * d:] == -7.0) and np.all(dp[npts:] == -7.0)
        tol = 1e-12
        assert np.abs(s[: npts * d].reshape(-1, d) - s_o).max() <= tol * np.abs(s_o).max()
        assert np.abs(dp[:npts] - dp_o).max() <= tol * max(np.abs(dp_o).max(), 1e-300)
        assert np.abs(C[: npts * d * d].reshape(-1, d, d) - C_o).max() <= tol * np.abs(C_o).max()
        ctx.set_option("host_chunk_points", old_chunk)
        t = [torch.from_numpy(a).cuda() for a in (u, sigma_n.reshape(-1), p)]
        Ct = torch.empty(npts * d * d, dtype=torch.float64, device="cuda")
        st = torch.empty(npts * d, dtype=torch.float64, device="cuda")
        dpt = torch.empty(npts, dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dm.von_mises(prm, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), Ct.data_ptr(), st.data_ptr(), dpt.data_ptr(),
                     mem=MEM_DEVICE)
        torch.cuda.synchronize()
        assert np.array_equal(Ct.cpu().numpy(), C[: npts * d * d]) and np.array_equal(st.cpu().numpy(), s[: npts * d])
        assert np.array_equal(dpt.cpu().numpy(), dp[:npts])
    finally:
        ctx.set_option("host_chunk_points", old_chunk)
        dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("with_tangent", [True, False])
def test_fused_von_mises_on_a_27_point_rule(ctx, oracle, with_tangent):
    """Q2 hexahedra with the 3x3x3 Gauss rule: tables 28.7 KB + four gather regions 28.7 KB leave no room for the landing slices of the
    global_load_lds prefetch (14 KB) inside 64 KB — the launch must take the same kernel with the loads in registers, not refuse the mesh
    (round-5 advisor finding; csrc/vm_field.hip field_launch)."""
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE, DeviceMesh, VmParams
    from tools.synthetic import gauss_tensor_rule, with_rule

    E = 70e3
    prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
    m = with_rule(structured_mesh("hexahedron", (5, 4, 3), 2, distort=0.2, seed=11), *gauss_tensor_rule("hexahedron", 3))
    assert m.nq == 27 and abs(m.weights.sum() - 1.0) < 1e-14
    d, npts = 6, m.num_cells * m.nq
    rng = np.random.Generator(np.random.PCG64(5))
    u = rng.normal(size=m.node_x.shape[0] * m.gdim)
    u *= 1.5e-3 / eval_operand(EPS_MANDEL, m.gdim, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi).std()
    sigma_n, p = _vm_state(npts, d, seed=2)
    e_ref = eval_operand(EPS_MANDEL, m.gdim, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
    C_o, s_o, dp_o = oracle.von_mises(e_ref.reshape(-1, d), sigma_n, p)
    assert 0.05 < (dp_o > 0).mean() < 0.98
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        t = [torch.from_numpy(a).cuda() for a in (u, sigma_n.reshape(-1), p)]
        Ct = torch.full((npts * d * d,), -7.0, dtype=torch.float64, device="cuda") if with_tangent else None
        st = torch.empty(npts * d, dtype=torch.float64, device="cuda")
        dpt = torch.empty(npts, dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        dm.von_mises(prm, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), Ct.data_ptr() if with_tangent else None, st.data_ptr(),
                     dpt.data_ptr(), mem=MEM_DEVICE)
        torch.cuda.synchronize()
        tol = 1e-12
        assert np.abs(st.cpu().numpy().reshape(-1, d) - s_o).max() <= tol * np.abs(s_o).max()
        assert np.abs(dpt.cpu().numpy() - dp_o).max() <= tol * np.abs(dp_o).max()
        if with_tangent:
            assert np.abs(Ct.cpu().numpy().reshape(-1, d, d) - C_o).max() <= tol * np.abs(C_o).max()
    finally:
        dm.close()


@pytest.mark.gpu
def test_lazy_operand_takes_the_fused_path_inside_the_reference_call_sequence(ctx, oracle):
    from dolfinx_external_operator_amd import (DeviceMesh, LazyOperand, QuadratureExternalOperator,
                                               evaluate_external_operators, evaluate_operands, make_von_mises)

    m = structured_mesh("triangle", (12, 9), 2, distort=0.2, seed=5)
    rng = np.random.Generator(np.random.PCG64(2))
    Du = rng.normal(0.0, 1e-4, size=m.node_x.shape[0] * 2)
    npts = m.num_cells * m.nq
    sigma_n, p = _vm_state(npts, 4, seed=3)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        deps = dm.operand("eps", Du, lazy=True)
        op = QuadratureExternalOperator(deps, num_cells=m.num_cells, num_points=m.nq, value_shape=(4, 4),
                                        external_function=make_von_mises(sigma_n, p, ctx=ctx), derivatives=(1,))
        ev = evaluate_operands([op])
        assert isinstance(ev[deps], LazyOperand) and ev[deps].shape == (m.num_cells, m.nq, 4)
        ((C_tang, sigma, dp),) = evaluate_external_operators([op], ev)
        assert ev[deps]._value is None                                   # the strain array was never materialised
        e_ref = eval_operand(EPS_MANDEL, 2, Du, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
        C_o, s_o, dp_o = oracle.von_mises(e_ref.reshape(-1, 4), sigma_n, p)
        assert np.abs(C_tang.reshape(-1, 4, 4) - C_o).max() <= 1e-12 * np.abs(C_o).max()
        assert np.abs(sigma.reshape(-1, 4) - s_o).max() <= 1e-12 * np.abs(s_o).max()
        assert np.array_equal(op.ref_coefficient.x.array, C_tang)
        # any other consumer sees an ordinary array
        assert np.abs(np.asarray(ev[deps]) - e_ref).max() <= 1e-13 * np.abs(e_ref).max()
        sub = deps.eval(np.array([3, 1], dtype=np.int32))               # entity subsets are evaluated eagerly
        assert isinstance(sub, np.ndarray) and np.abs(sub - e_ref[[3, 1]]).max() <= 1e-13 * np.abs(e_ref).max()
    finally:
        dm.close()


@pytest.mark.gpu
def test_load_stepping_example_runs():
    """examples/von_mises_load_stepping.py: the reference's calling sequence with the lazy operand + fused kernel."""
    import importlib.util
    import pathlib

    path = pathlib.Path(__file__).resolve().parents[1] / "examples" / "von_mises_load_stepping.py"
    spec = importlib.util.spec_from_file_location("vm_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(24)
    fr = [s["plastic_fraction"] for s in rep["steps"]]
    assert fr[0] < fr[2] and fr[3] == 0.0                     # yielding spreads under loading, unloading is elastic
    assert rep["steps"][2]["max_p"] > 0
    res = mod.main(24, "resident")                            # history variables in the device mirror: the same trajectory, bit for bit
    assert np.array_equal(res["final_p"], rep["final_p"]) and np.array_equal(res["final_sigma_n"], rep["final_sigma_n"])


@pytest.mark.gpu
@pytest.mark.parametrize("cell,n", [("triangle", (9, 7)), ("hexahedron", (3, 3, 2)), ("quadrilateral", (5, 5))])
def test_value_and_gradient_in_one_pass_and_fused_heat(ctx, oracle, cell, n):
    """Operand kind "value_grad" against the oracle, and dxo_heat_field (T, grad T and the heat-flux kernels of
    demo_nonlinear_heat_equation_part2.py:219-261 in one launch) against operand oracle -> heat oracle; then the same
    through make_heat with two lazy operands of one field, as the demo's three operators would call it."""
    from dolfinx_external_operator_amd import DeviceMesh, make_heat
    from oracle.operand_oracle import VALUE_GRAD

    m = structured_mesh(cell, n, 2, distort=0.2, seed=2)
    G = m.gdim
    x = m.node_x
    Tn = 1.0 + x[:, 0] ** 2 + x[:, 1] + (0.3 * x[:, 2] * x[:, 0] if G == 3 else 0.0)     # T = x^2 + y (:148) is in the space
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        vg = dm.evaluate("value_grad", 1, Tn)
        ref = eval_operand(VALUE_GRAD, 1, Tn, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi)
        assert vg.shape == (m.num_cells, m.nq, 1 + G)
        assert np.abs(vg - ref).max() <= 1e-13 * np.abs(ref).max()
        xq = m.physical_points()
        assert np.abs(vg[..., 0] - (1.0 + xq[..., 0] ** 2 + xq[..., 1] + (0.3 * xq[..., 2] * xq[..., 0] if G == 3 else 0.0))).max() <= 1e-12
        npts = m.num_cells * m.nq
        q, dT, ds = np.full(npts * G + 2, -3.0), np.full(npts * G + 2, -3.0), np.full(npts * G * G + 2, -3.0)
        dm.heat(1.0, 1.0, Tn, q, dT, ds)
        assert q[-1] == -3.0 and dT[-2] == -3.0 and ds[-1] == -3.0
        if G == 2:
            qo, dTo, dso = oracle.heat(ref[..., 0].reshape(-1), ref[..., 1:].reshape(-1, 2))
        else:   # the C oracle is the 2-D reference kernel; 3-D: the same three formulas in NumPy
            k = 1.0 / (1.0 + ref[..., 0].reshape(-1))
            sg = ref[..., 1:].reshape(-1, 3)
            qo, dTo, dso = -k[:, None] * sg, (k * k)[:, None] * sg, -k[:, None, None] * np.eye(3)[None]
        for got, want in ((q[:-2], qo), (dT[:-2], dTo), (ds[:-2], dso)):
            assert np.abs(got - np.asarray(want).reshape(-1)).max() <= 1e-13 * np.abs(want).max()
        only = np.empty(npts * G)
        dm.heat(1.0, 1.0, Tn, None, only, None)                      # a single requested output
        assert np.array_equal(only, dT[:-2])
        # the demo's call pattern: q((0,0)), dq/dT((1,0)), dq/dsigma((0,1)) on (T, sigma) operands of the same field
        T_op, s_op = dm.operand("value", Tn, bs=1, lazy=True), dm.operand("grad", Tn, bs=1, lazy=True)
        ext = make_heat(A=1.0, B=1.0, ctx=ctx)
        Tv, sv = T_op.eval(None), s_op.eval(None)
        for deriv, want in (((0, 0), q[:-2]), ((1, 0), dT[:-2]), ((0, 1), ds[:-2])):
            assert np.array_equal(ext(deriv)(Tv, sv), want)
        assert Tv._value is None and sv._value is None               # neither operand array was ever materialised
    finally:
        dm.close()


# ------------------------------------------------------------------------------------------------ codim-1 entities
def _facet_case(cell, degree, seed):
    from tools.synthetic import FACETS, facet_physical_points, facet_tables

    m = structured_mesh(cell, CELLS[cell], degree, distort=0.0 if degree == 2 and cell in ("quadrilateral", "hexahedron") else 0.25, seed=seed)
    phi_f, dphi_f, dpsi_f, ref_pts = facet_tables(m)
    rng = np.random.Generator(np.random.PCG64(seed))
    n = 3 * m.num_cells
    ents = np.stack([rng.integers(0, m.num_cells, n), rng.integers(0, len(FACETS[cell]), n)], axis=1).astype(np.int32)
    return m, (phi_f, dphi_f, dpsi_f), ents, facet_physical_points(m, ents, ref_pts)


@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_oracle_on_cell_facet_pairs(cell, degree):
    """(cell, local_facet) entities as evaluate_operands hands them to Expression.eval (external_operator.py:340, 402;
    test/test_codim_external_operator.py:76-84): polynomial fields of the element's degree are reproduced at the facet
    points, with the FULL physical gradient. (Degree 2 on tensor cells needs undistorted cells to be exact.)"""
    from oracle.operand_oracle import eval_operand_facets

    m, (phi_f, dphi_f, dpsi_f), ents, xq = _facet_case(cell, degree, seed=5)
    for bs in (1, m.gdim):
        u, grad = poly_field(m.gdim, bs, degree, seed=20 + bs)
        uvec = u(m.node_x).reshape(-1)
        for kind in ("value", "grad") + (("eps", "F", "C", "I1", "detF", "div") if bs == m.gdim else ()):
            got = eval_operand_facets(KIND_ID[kind], bs, uvec, m.dofmap, m.geom_dofmap, m.x, phi_f, dphi_f, dpsi_f, ents)
            want = expected(kind, grad(xq), u(xq))
            assert got.shape == want.shape == (len(ents), phi_f.shape[1], want.shape[2])
            assert np.abs(got - want).max() <= 5e-12 * max(1.0, np.abs(want).max()), (cell, degree, bs, kind)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", list(CELLS))
@pytest.mark.parametrize("degree", [1, 2])
def test_hip_on_cell_facet_pairs(ctx, cell, degree):
    from dolfinx_external_operator_amd import DeviceMesh, Operand, QuadratureExternalOperator, evaluate_operands
    from oracle.operand_oracle import eval_operand_facets

    m, tabs, ents, xq = _facet_case(cell, degree, seed=6)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        with pytest.raises(ValueError, match="dxo_mesh_set_facet_tables"):
            dm.evaluate_facets("value", 1, np.zeros(m.node_x.shape[0]), ents)
        dm.set_facet_tables(*tabs)
        rng = np.random.Generator(np.random.PCG64(3))
        for bs in (1, m.gdim):
            u, grad = poly_field(m.gdim, bs, degree, seed=30 + bs)
            uvec = u(m.node_x).reshape(-1)
            rough = rng.normal(size=uvec.size)
            for kind in ("value", "grad") + (("eps", "F", "C", "I1", "detF", "div") if bs == m.gdim else ()):
                got = dm.evaluate_facets(kind, bs, uvec, ents)
                want = expected(kind, grad(xq), u(xq))
                assert np.abs(got - want).max() <= 5e-12 * max(1.0, np.abs(want).max()), (cell, degree, bs, kind)
                ref = eval_operand_facets(KIND_ID[kind], bs, rough, m.dofmap, m.geom_dofmap, m.x, *tabs, ents)
                got = dm.evaluate_facets(kind, bs, rough, ents)
                assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (cell, degree, bs, kind)
        # through the dispatcher: a 2-D entity array reaches the operand's eval unchanged and comes back (n, nq_f, ...)
        field = rng.normal(size=m.node_x.shape[0])
        operand = dm.operand("value", field, bs=1)
        op = QuadratureExternalOperator(operand, num_cells=len(ents), num_points=tabs[0].shape[1], value_shape=(),
                                        external_function=lambda d: (lambda x: np.cos(x).reshape(-1)))
        table = evaluate_operands([op], entities=ents)
        ref = eval_operand_facets(VALUE, 1, field, m.dofmap, m.geom_dofmap, m.x, *tabs, ents)[..., 0]
        assert table[operand].shape == ref.shape and np.abs(table[operand] - ref).max() <= 1e-13 * np.abs(ref).max()
        bad = ents.copy()
        bad[0, 1] = 7
        with pytest.raises(ValueError, match="entity outside"):
            dm.evaluate_facets("value", 1, field, bad)
        assert dm.evaluate_facets("value", 1, field, np.empty((0, 2), dtype=np.int32)).shape == (0, tabs[0].shape[1], 1)
    finally:
        dm.close()


@pytest.mark.gpu
def test_lazy_operand_snapshot_or_live_field(ctx):
    """`operand(..., lazy=True)` snapshots the field vector at evaluate_operands time (what Expression.eval's result is);
    `snapshot=False` keeps the live array for the reference's back-to-back calling sequence — no copy of the dof vector."""
    from dolfinx_external_operator_amd import DeviceMesh

    m = structured_mesh("triangle", (6, 5), 2)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    u = np.linspace(0.0, 1.0, m.node_x.shape[0] * 2)
    snap = dm.operand("eps", u, lazy=True).eval(None)
    live = dm.operand("eps", u, lazy=True, snapshot=False).eval(None)
    assert not np.shares_memory(snap.u, u) and np.shares_memory(live.u, u)
    before = np.asarray(dm.operand("eps", u).eval(None))
    u *= 2.0                                                    # the field changes after "evaluation"
    assert np.array_equal(np.asarray(snap), before)             # the snapshot still is the old value
    assert np.max(np.abs(np.asarray(live) - 2.0 * before)) <= 1e-14 * np.max(np.abs(before))   # the live operand follows the field
    dm.close()


@pytest.mark.gpu
def test_the_reference_operand_test_pattern_on_the_device(ctx):
    """test/test_operands_evaluation.py:19-66 restated on the device: N = FEMExternalOperator(I1, slope) with I1 = tr(F.T * F),
    F = Identity(d) + grad(u) of a P1 vector field u = (0.1 x, 0.3 y), and `slope` a P1 scalar function equal to 1: evaluate_operands
    must return I1 at every quadrature point (known answer: F = diag(1.1, 1.3), I1 = 2.9, det F = 1.43, C = diag(1.21, 1.69)) and the
    slope's values, with the reference's shapes (scalar operands are 2-D: (num_cells, nq)). Nonlinear operands are forward-only: the
    adjoint refuses them."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh, QuadratureExternalOperator, evaluate_operands

    m = structured_mesh("triangle", (4, 4), 1, distort=0.2, seed=2)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        u = np.stack([0.1 * m.node_x[:, 0], 0.3 * m.node_x[:, 1]], axis=1).reshape(-1)        # :20
        slope = np.ones(m.node_x.shape[0])                                                      # :39
        I1_op, slope_op = dm.operand("I1", u), dm.operand("value", slope, bs=1)
        N = QuadratureExternalOperator(I1_op, slope_op, num_cells=m.num_cells, num_points=m.nq, value_shape=(),
                                       external_function=lambda d: (lambda a, b: (a * b).reshape(-1)))
        ev = evaluate_operands([N])
        assert ev[I1_op].shape == ev[slope_op].shape == (m.num_cells, m.nq)
        np.testing.assert_allclose(ev[I1_op], 1.1 ** 2 + 1.3 ** 2, rtol=1e-13)                  # :65
        np.testing.assert_allclose(ev[slope_op], 1.0, rtol=1e-13)                               # :66
        np.testing.assert_allclose(dm.operand("detF", u).eval(None), 1.1 * 1.3, rtol=1e-13)
        C = dm.operand("C", u).eval(None)
        assert C.shape == (m.num_cells, m.nq, 2, 2)
        np.testing.assert_allclose(C, np.broadcast_to(np.diag([1.21, 1.69]), C.shape), atol=1e-13)
        S = torch.zeros(m.num_cells * m.nq, dtype=torch.float64, device="cuda")
        out = torch.zeros(m.node_x.shape[0] * 2, dtype=torch.float64, device="cuda")
        with pytest.raises(ValueError, match="nonlinear operand"):
            dm.adjoint("I1", 2, S.data_ptr(), out.data_ptr())
        with pytest.raises(ValueError):
            dm.value_size("I1", 1)                                                              # needs a vector field, bs = gdim
    finally:
        dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cell,degree,bs", [("triangle", 1, 4), ("triangle", 2, 3), ("quadrilateral", 2, 5), ("tetrahedron", 1, 4), ("tetrahedron", 2, 2),
                                            ("hexahedron", 2, 4), ("hexahedron", 1, 9)])
def test_fields_of_any_block_size_as_value_and_gradient_operands(ctx, cell, degree, bs):
    """`evaluate_operands` evaluates whatever field the operand is (external_operator.py:386-402): test/test_nested_ex_op.py:113-118 hands a
    4-component DG field `theta` to an operator. value / grad / value_grad of a field whose block size is neither 1 nor gdim run as one scalar
    launch per component (csrc/operand.hip dispatch_components): against the NumPy oracle on all cells, on an entity list, on (cell, facet) pairs,
    and with Expression.eval's shapes through the dispatcher."""
    from dolfinx_external_operator_amd import DeviceMesh, QuadratureExternalOperator, evaluate_operands
    from oracle.operand_oracle import eval_operand_facets

    m = structured_mesh(cell, CELLS[cell], degree, distort=0.15, seed=4)
    g = m.gdim
    assert bs not in (1, g)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    rng = np.random.Generator(np.random.PCG64(40 + bs))
    u = rng.normal(size=m.node_x.shape[0] * bs)
    ents = rng.permutation(m.num_cells)[: max(3, m.num_cells // 2)].astype(np.int32)
    try:
        for kind in ("value", "grad", "value_grad"):
            assert dm.value_size(kind, bs) == {"value": bs, "grad": bs * g, "value_grad": bs * (1 + g)}[kind]
            for cells in (None, ents):
                ref = eval_operand(KIND_ID[kind], bs, u, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, cells)
                got = dm.evaluate(kind, bs, u, cells)
                assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), (kind, cells is None)
        # each component is the scalar operand of that component (the dense bs = 1 kernel): the same bits
        one = dm.evaluate("grad", 1, np.ascontiguousarray(u.reshape(-1, bs)[:, bs - 1]))
        assert np.array_equal(dm.evaluate("grad", bs, u).reshape(m.num_cells, m.nq, bs, g)[:, :, bs - 1, :], one)
        # Expression.eval's shapes through the dispatcher: (cells, nq, bs) and (cells, nq, bs, gdim)
        theta, dtheta = dm.operand("value", u, bs=bs), dm.operand("grad", u, bs=bs)
        N = QuadratureExternalOperator(theta, dtheta, num_cells=m.num_cells, num_points=m.nq, value_shape=(),
                                       external_function=lambda d: (lambda a, b: (a.sum(axis=2) + b.sum(axis=(2, 3))).reshape(-1)))
        ev = evaluate_operands([N])
        assert ev[theta].shape == (m.num_cells, m.nq, bs) and ev[dtheta].shape == (m.num_cells, m.nq, bs, g)
    finally:
        dm.close()
    mf, tabs, fents, _ = _facet_case(cell, degree, seed=6)
    dmf = DeviceMesh.from_synthetic(mf, ctx=ctx)
    try:
        dmf.set_facet_tables(*tabs)
        uf = rng.normal(size=mf.node_x.shape[0] * bs)
        for kind in ("value", "grad", "value_grad"):
            ref = eval_operand_facets(KIND_ID[kind], bs, uf, mf.dofmap, mf.geom_dofmap, mf.x, *tabs, fents)
            got = dmf.evaluate_facets(kind, bs, uf, fents)
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max(), kind
        # the kinds built from a displacement gradient still need bs = gdim, and the adjoint takes 1 or gdim
        with pytest.raises(ValueError):
            dmf.value_size("eps", bs)
        with pytest.raises(ValueError):
            dmf.value_size("value", 65)
    finally:
        dmf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cell", list(CELLS))
def test_the_operands_of_the_nested_operator_test_x_and_a_four_component_field(ctx, cell):
    """test/test_nested_ex_op.py:113-130: N = FEMExternalOperator(x, theta, ...) with x = ufl.SpatialCoordinate(mesh) and theta a 4-component
    DG field, the kernel u_NN_impl(gdim, x, theta) of :60-83 reading x[..., j] and theta[..., k]. Both operands on the device: x against the
    coordinate map of the synthetic mesh (distorted, so tensor cells are genuinely non-affine), theta = 0.32 as in :109, shapes
    (cells, nq, gdim) and (cells, nq, 4), all cells and an entity list; the dispatcher deduplicates x when two operators share it (:383-399)."""
    from dolfinx_external_operator_amd import DeviceMesh, QuadratureExternalOperator, evaluate_operands

    m = structured_mesh(cell, CELLS[cell], 1, distort=0.2, seed=9)
    g = m.gdim
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    try:
        xq = m.physical_points()                                             # (cells, nq, gdim), NumPy
        got = dm.coordinate()
        assert got.shape == xq.shape and np.abs(got - xq).max() <= 1e-14 * np.abs(xq).max()
        ents = np.arange(m.num_cells - 1, -1, -2, dtype=np.int32)
        assert np.array_equal(dm.coordinate(ents), got[ents])
        theta = np.full(m.node_x.shape[0] * 4, 0.32)                         # :109
        x_op, th_op = dm.operand("x", None), dm.operand("value", theta, bs=4)

        def u_nn(derivatives):                                               # the shape logic of :60-83: one value per point from x and theta
            return lambda x, th: (np.tanh(x[..., 0] * th[..., 0] + th[..., 1]) * th[..., 2] + th[..., 3] * x[..., g - 1]).reshape(-1)

        N1 = QuadratureExternalOperator(x_op, th_op, num_cells=m.num_cells, num_points=m.nq, value_shape=(), external_function=u_nn)
        N2 = QuadratureExternalOperator(x_op, num_cells=m.num_cells, num_points=m.nq, value_shape=(), external_function=lambda d: (lambda x: x[..., 0].reshape(-1)))
        ev = evaluate_operands([N1, N2])
        assert x_op.eval_count == 1                                          # evaluated once for both operators
        assert ev[x_op].shape == (m.num_cells, m.nq, g) and ev[th_op].shape == (m.num_cells, m.nq, 4)
        np.testing.assert_allclose(ev[th_op], 0.32, rtol=1e-14)
        want = np.tanh(xq[..., 0] * 0.32 + 0.32) * 0.32 + 0.32 * xq[..., g - 1]
        from dolfinx_external_operator_amd import evaluate_external_operators
        vals = evaluate_external_operators([N1, N2], ev)
        np.testing.assert_allclose(N1.ref_coefficient.x.array.reshape(m.num_cells, m.nq), want, rtol=1e-13)
        np.testing.assert_allclose(N2.ref_coefficient.x.array.reshape(m.num_cells, m.nq), xq[..., 0], rtol=1e-13, atol=1e-15)
        assert len(vals) == 2
        bare = DeviceMesh(gdim=g, phi=m.phi, dphi=m.dphi, dpsi=m.dpsi, dofmap=m.dofmap, geom_dofmap=m.geom_dofmap, x=m.x,
                          num_field_nodes=m.node_x.shape[0], ctx=ctx)
        try:
            with pytest.raises(ValueError, match="dxo_mesh_set_coordinate_values"):
                bare.coordinate()
        finally:
            bare.close()
    finally:
        dm.close()
