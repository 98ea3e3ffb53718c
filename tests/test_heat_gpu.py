"""Parity of the HIP heat-flux kernel with the reference golden (config 1) and the oracle."""
import numpy as np
import pytest

from dolfinx_external_operator_amd import (
    MEM_DEVICE,
    MEM_HOST,
    Operand,
    QuadratureExternalOperator,
    evaluate_external_operators,
    evaluate_operands,
    make_heat,
)

pytestmark = pytest.mark.gpu

# k = 1/(A + B T) uses the GPU's correctly-rounded fp64 division; products may be FMA-contracted.
RTOL = 1e-15


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_config1_golden_through_drop_in_callback(ctx, golden):
    g = golden("heat_c1.npz")
    T_full, s_full = g["T"], g["sigma"]
    nc, nq = T_full.shape
    q_external = make_heat(A=float(g["A"]), B=float(g["B"]), ctx=ctx)
    T = Operand(lambda cells: T_full[cells], "T")
    sigma = Operand(lambda cells: s_full[cells].reshape(len(cells), -1), "grad(T)")  # (nc, nq*gdim), part2.py:222
    q = QuadratureExternalOperator(T, sigma, num_cells=nc, num_points=nq, value_shape=(2,), external_function=q_external)
    dqdT = QuadratureExternalOperator(T, sigma, num_cells=nc, num_points=nq, value_shape=(2,),
                                      external_function=q_external, derivatives=(1, 0))
    dqds = QuadratureExternalOperator(T, sigma, num_cells=nc, num_points=nq, value_shape=(2, 2),
                                      external_function=q_external, derivatives=(0, 1))
    ev = evaluate_operands([q])
    evaluate_external_operators([q], ev)            # part2.py:306-309
    evaluate_external_operators([dqdT, dqds], ev)
    assert rel(q.ref_coefficient.x.array, g["q"]) <= RTOL
    assert rel(dqdT.ref_coefficient.x.array, g["dqdT"]) <= RTOL
    assert rel(dqds.ref_coefficient.x.array, g["dqdsigma"]) <= RTOL
    assert np.array_equal(dqds.ref_coefficient.x.array == 0, g["dqdsigma"] == 0)
    assert np.allclose(q.ref_coefficient.x.array, g["q"])       # the reference's own criterion (part2.py:320)
    with pytest.raises(NotImplementedError):
        q_external((1, 1))


def test_fused_by_identity_gives_same_values(ctx, golden):
    g = golden("heat_c1.npz")
    T, s = g["T"], g["sigma"]
    fused = make_heat(ctx=ctx, fuse_by_identity=True)
    plain = make_heat(ctx=ctx, fuse_by_identity=False)
    for idx in ((0, 0), (1, 0), (0, 1)):
        assert np.array_equal(fused(idx)(T, s), plain(idx)(T, s))


def test_default_fuses_one_pass_of_numpy_operands_and_the_tripwire_catches_in_place_updates(ctx, golden):
    """Round 6 default (fuse_by_identity=None): the three derivative calls of one pass on the SAME NumPy operand objects are one
    launch (part2.py:307-309 drives them from one evaluated_operands dict); fresh arrays, other shapes or an in-place update of an
    operand (caught by the 48-entry tripwire) launch again, and the values are those of the unfused calls bit for bit."""
    g = golden("heat_c1.npz")
    T, s = g["T"].copy(), g["sigma"].copy()
    q_external = make_heat(A=float(g["A"]), B=float(g["B"]), ctx=ctx)
    plain = make_heat(A=float(g["A"]), B=float(g["B"]), ctx=ctx, fuse_by_identity=False)
    launches = []
    real = ctx.heat
    ctx.heat = lambda *a, **k: (launches.append(1), real(*a, **k))[1]
    try:
        got = [q_external(idx)(T, s) for idx in ((0, 0), (1, 0), (0, 1))]
        assert len(launches) == 1
        T2, s2 = T.copy(), s.copy()                      # the next pass: fresh operand arrays
        got2 = [q_external(idx)(T2, s2) for idx in ((0, 0), (1, 0), (0, 1))]
        assert len(launches) == 2
        T2 *= 1.5                                        # a new solution written IN PLACE between two derivative calls
        stale_guard = q_external((1, 0))(T2, s2)
        assert len(launches) == 3
    finally:
        ctx.heat = real
    for k, idx in enumerate(((0, 0), (1, 0), (0, 1))):
        ref = plain(idx)(T, s)
        assert np.array_equal(got[k], ref) and np.array_equal(got2[k], ref)
    assert np.array_equal(stale_guard, plain((1, 0))(T2, s2)) and not np.array_equal(stale_guard, got2[1])
    assert rel(got[0], g["q"]) <= RTOL and rel(got[2], g["dqdsigma"]) <= RTOL


def test_bind_fuses_one_pass_and_writes_the_three_coefficients_in_place(ctx, golden):
    """q_external.bind(q, dqdT, dqdsigma): one launch per evaluate_external_operators pass fills all three coefficient arrays
    (same bits as the unbound calls), the dispatcher's assignment is array-to-itself, and fresh operand arrays (the next
    pass's evaluate_operands) trigger a new launch."""
    g = golden("heat_c1.npz")
    T_full, s_full = g["T"], g["sigma"]
    nc, nq = T_full.shape
    scale = {"v": 1.0}
    T = Operand(lambda cells: T_full[cells] * scale["v"], "T")
    sigma = Operand(lambda cells: s_full[cells].reshape(len(cells), -1), "grad(T)")
    q_external = make_heat(A=float(g["A"]), B=float(g["B"]), ctx=ctx)
    ops = [QuadratureExternalOperator(T, sigma, num_cells=nc, num_points=nq, value_shape=shape, external_function=q_external, derivatives=dv)
           for shape, dv in (((2,), (0, 0)), ((2,), (1, 0)), ((2, 2), (0, 1)))]
    assert q_external.bind(*ops) is q_external
    launches = []
    real = ctx.heat
    ctx.heat = lambda *a, **k: (launches.append(1), real(*a, **k))[1]
    try:
        ev = evaluate_operands(ops)
        res = evaluate_external_operators(ops, ev)
        assert len(launches) == 1
        for op, r, key in zip(ops, res, ("q", "dqdT", "dqdsigma")):
            assert np.shares_memory(r, op.ref_coefficient.x.array)
            assert rel(op.ref_coefficient.x.array, g[key]) <= RTOL
        scale["v"] = 1.5                                    # a new Newton iterate: evaluate_operands returns new arrays
        ev = evaluate_operands(ops)
        evaluate_external_operators(ops, ev)
        assert len(launches) == 2
    finally:
        ctx.heat = real
    plain = make_heat(A=float(g["A"]), B=float(g["B"]), ctx=ctx)
    for op, dv in zip(ops, ((0, 0), (1, 0), (0, 1))):
        assert np.array_equal(op.ref_coefficient.x.array, plain(dv)(T_full * 1.5, s_full))
    with pytest.raises(ValueError):
        make_heat(ctx=ctx).bind(ops[2], None, None)((0, 0))(T_full, s_full)     # wrong coefficient size for q


@pytest.mark.parametrize("gdim", [1, 2, 3])
@pytest.mark.parametrize("n", [0, 1, 63, 65, 1000, 6144])
def test_sizes_and_dims_against_oracle(ctx, oracle, gdim, n):
    rng = np.random.default_rng(n + gdim)
    T = rng.uniform(0.1, 3.0, size=n)
    s = rng.normal(size=(n, gdim))
    q, dT, ds = np.empty(n * gdim), np.empty(n * gdim), np.empty(n * gdim * gdim)
    ctx.heat(1.3, 0.7, gdim, n, MEM_HOST, T, s, q, dT, ds)
    if n == 0:
        return
    qo, dTo, dso = oracle.heat(T, s, A=1.3, B=0.7, gdim=gdim)
    assert rel(q, qo.reshape(-1)) <= RTOL and rel(dT, dTo.reshape(-1)) <= RTOL and rel(ds, dso.reshape(-1)) <= RTOL
    # partial output selection
    q2 = np.empty(n * gdim)
    ctx.heat(1.3, 0.7, gdim, n, MEM_HOST, T, s, q2, None, None)
    assert np.array_equal(q2, q)


def test_device_path_large(ctx, oracle):
    import torch

    n = 1_000_003
    rng = np.random.default_rng(1)
    T = rng.uniform(0.1, 3.0, size=n)
    s = rng.normal(size=(n, 2))
    dev = torch.device("cuda:0")
    Tt, st = torch.from_numpy(T).to(dev), torch.from_numpy(s).to(dev)
    q = torch.empty(n * 2, dtype=torch.float64, device=dev)
    dT = torch.empty(n * 2, dtype=torch.float64, device=dev)
    ds = torch.full((n * 4 + 8,), -7.0, dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, Tt.data_ptr(), st.data_ptr(), q.data_ptr(), dT.data_ptr(), ds.data_ptr())
    torch.cuda.synchronize()
    qo, dTo, dso = oracle.heat(T, s, nthreads=8)
    assert rel(q.cpu().numpy(), qo.reshape(-1)) <= RTOL
    assert rel(dT.cpu().numpy(), dTo.reshape(-1)) <= RTOL
    assert rel(ds[: n * 4].cpu().numpy(), dso.reshape(-1)) <= RTOL
    assert torch.all(ds[n * 4:] == -7.0)
