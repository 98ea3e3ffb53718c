"""Parity of the HIP Mohr-Coulomb kernel (through the C ABI) with the oracle and the reference-source goldens."""
import pathlib

import numpy as np
import pytest

from conftest import mc_compare_all, mc_compare, mc_elastic_matrices, mc_tracing_inputs
from dolfinx_external_operator_amd import (
    MEM_DEVICE,
    MEM_HOST,
    McParams,
    Operand,
    QuadratureExternalOperator,
    evaluate_external_operators,
    evaluate_operands,
    make_mohr_coulomb,
)

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).resolve().parent / "golden" / "mohr_coulomb.npz"


def params(**kw):
    d = dict(E=6778.0, nu=0.25, c=3.45, phi=np.pi / 6, psi=np.pi / 6, theta_T=26 * np.pi / 180, a=None, tol=1e-8, nitermax=200)
    d.update(kw)
    if d["a"] is None:
        d["a"] = 0.26 * d["c"] / np.tan(d["phi"])
    return McParams(d["E"], d["nu"], d["c"], d["phi"], d["psi"], d["theta_T"], d["a"], d["tol"], d["nitermax"], 0)


def run_host(ctx, deps, sn, diag=True, **kw):
    n = len(deps)
    Ct, s = np.empty(n * 16), np.empty(n * 4)
    it = np.empty(n, dtype=np.int32) if diag else None
    y, nr, dl = (np.empty(n), np.empty(n), np.empty(n)) if diag else (None, None, None)
    ctx.mohr_coulomb(params(**kw), n, MEM_HOST, np.ascontiguousarray(deps), np.ascontiguousarray(sn), Ct, s, it, y, nr, dl)
    return Ct.reshape(n, 4, 4), s.reshape(n, 4), it, y, nr, dl


def run_device(ctx, deps, sn, **kw):
    import torch

    n = len(deps)
    dev = torch.device("cuda:0")
    d, s0 = torch.from_numpy(np.ascontiguousarray(deps)).to(dev), torch.from_numpy(np.ascontiguousarray(sn)).to(dev)
    Ct = torch.empty(n * 16, dtype=torch.float64, device=dev)
    s = torch.empty(n * 4, dtype=torch.float64, device=dev)
    it = torch.empty(n, dtype=torch.int32, device=dev)
    y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.mohr_coulomb(params(**kw), n, MEM_DEVICE, d.data_ptr(), s0.data_ptr(), Ct.data_ptr(), s.data_ptr(), it.data_ptr(),
                     y.data_ptr(), nr.data_ptr(), dl.data_ptr())
    torch.cuda.synchronize()
    return (Ct.cpu().numpy().reshape(n, 4, 4), s.cpu().numpy().reshape(n, 4), it.cpu().numpy(), y.cpu().numpy(),
            nr.cpu().numpy(), dl.cpu().numpy())


@pytest.mark.skipif(not GOLD.exists(), reason="golden not generated")
def test_reference_source_golden(ctx):
    g = np.load(GOLD)
    prm = {k[4:]: g[k].item() for k in g.files if k.startswith("prm_")}
    prm["nitermax"] = int(prm["nitermax"])
    ref = (g["C_tang"], g["sigma"], g["niter"], g["yielding"], g["norm_res"], g["dlambda"])
    for runner in (run_host, run_device):
        got = runner(ctx, g["deps"], g["sigma_n"], **prm)
        mc_compare(got, ref, f"HIP {runner.__name__} vs reference golden", g["sigma_n"])
    zero = np.flatnonzero(g["tag"] == -2)[0]
    assert got[2][zero] == 0 and np.all(got[0][zero] == 0.0)


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 20000])
def test_tracing_distribution_against_oracle(ctx, oracle, n, variant):
    deps, sn = mc_tracing_inputs(oracle, n, seed=50 + n)
    ctx.set_option("mc_variant", variant)
    try:
        got = run_device(ctx, deps, sn)
    finally:
        ctx.set_option("mc_variant", 2)
    if n == 0:
        assert got[0].size == 0
        return
    ref = oracle.mohr_coulomb(deps, sn, nthreads=8)
    mc_compare(got, ref, "HIP vs oracle", sn)


def test_shear_and_non_associated_flow(ctx, oracle):
    deps, sn = mc_tracing_inputs(oracle, 5000, seed=6, shear=0.3)
    for kw in ({}, {"psi": 20 * np.pi / 180}, {"phi": 25 * np.pi / 180, "psi": 10 * np.pi / 180, "theta_T": 20 * np.pi / 180}):
        ref = oracle.mohr_coulomb(deps, sn, nthreads=8, **kw)
        got = run_device(ctx, deps, sn, **kw)
        assert (ref[2] < 30).mean() > 0.95
        # every point is compared: fast ones to the tight bounds, slowly converging ones on sigma / norm_res, and the
        # set of non-converged points (reported through niter == Nitermax, never raised) must be the oracle's
        mc_compare_all(got, ref, f"HIP vs oracle {kw}", sn)


def test_outputs_without_diagnostics_and_guards(ctx, oracle):
    import torch

    n = 777
    deps, sn = mc_tracing_inputs(oracle, n, seed=3)
    dev = torch.device("cuda:0")
    d, s0 = torch.from_numpy(deps).to(dev), torch.from_numpy(sn).to(dev)
    Ct = torch.full((n * 16 + 32,), -7.0, dtype=torch.float64, device=dev)
    s = torch.full((n * 4 + 32,), -7.0, dtype=torch.float64, device=dev)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.mohr_coulomb(params(), n, MEM_DEVICE, d.data_ptr(), s0.data_ptr(), Ct.data_ptr(), s.data_ptr())
    torch.cuda.synchronize()
    assert torch.all(Ct[n * 16:] == -7.0) and torch.all(s[n * 4:] == -7.0)
    ref = oracle.mohr_coulomb(deps, sn, nthreads=8)
    assert np.max(np.abs(s[: n * 4].cpu().numpy() - ref[1].reshape(-1))) < 1e-11
    full = run_host(ctx, deps, sn, diag=False)
    assert np.max(np.abs(full[1] - ref[1])) < 1e-11 and full[2] is None


def test_argument_validation(ctx):
    a = np.zeros(64)
    with pytest.raises(ValueError, match="SIZE"):
        ctx.mohr_coulomb(params(), -1, MEM_HOST, a, a, a, a)
    with pytest.raises(ValueError, match="NULL"):
        ctx.mohr_coulomb(params(), 1, MEM_HOST, a, None, a, a)
    with pytest.raises(ValueError, match="MEM"):
        ctx.mohr_coulomb(params(), 1, 5, a, a, a, a)


def test_drop_in_external_function(ctx, oracle):
    """The demo's call sequence (demo_plasticity_mohr_coulomb.py:679-688) with the HIP-backed callback."""
    nc, nq = 300, 3
    deps, sn = mc_tracing_inputs(oracle, nc * nq, seed=12)
    state = {"sigma_n": sn.reshape(-1).copy()}
    seen = []
    sigma_external = make_mohr_coulomb(lambda: state["sigma_n"], ctx=ctx, on_summary=seen.append)
    deps_full = deps.reshape(nc, nq, 4)
    eps = Operand(lambda cells: deps_full[cells], "eps(Du)")
    sigma = QuadratureExternalOperator(eps, num_cells=nc, num_points=nq, value_shape=(4,), external_function=sigma_external)
    C_tang = QuadratureExternalOperator(eps, num_cells=nc, num_points=nq, value_shape=(4, 4),
                                        external_function=sigma_external, derivatives=(1,))
    evaluated_operands = evaluate_operands([sigma])
    ((_, sigma_new),) = evaluate_external_operators([C_tang], evaluated_operands)   # :686
    sigma.ref_coefficient.x.array[:] = sigma_new                                     # :688
    ref = oracle.mohr_coulomb(deps, sn, nthreads=8)
    niter, yielding, norm_res, dlambda = sigma_external.last_state
    got = (C_tang.ref_coefficient.x.array.reshape(-1, 4, 4), sigma.ref_coefficient.x.array.reshape(-1, 4), niter, yielding,
           norm_res, dlambda)
    mc_compare(got, ref, "drop-in vs oracle", sn)
    u, c = np.unique(ref[2], return_counts=True)                                     # the reference's printed summary (:584-591)
    assert np.array_equal(seen[0]["unique_iters"], u) and np.array_equal(seen[0]["counts"], c)
    assert abs(seen[0]["max_yielding"] - ref[3].max()) < 1e-12
    with pytest.raises(NotImplementedError, match="No external function is defined"):
        evaluate_external_operators([sigma], evaluated_operands)


def test_kernel_variants_agree_bitwise(ctx, oracle):
    """Lane = point (0), classify + compacted Newton (1) and the single persistent kernel (2) schedule the SAME per-point
    arithmetic (mc_core.h lane_pass): every output is bit-identical, whatever the order the points were taken in."""
    deps, sn = mc_tracing_inputs(oracle, 150_001, seed=8, shear=0.2)
    out = []
    try:
        for variant in (0, 1, 2):
            ctx.set_option("mc_variant", variant)
            out.append(run_device(ctx, deps, sn))
    finally:
        ctx.set_option("mc_variant", 2)
    for b in out[1:]:
        assert np.array_equal(out[0][2], b[2])                  # iteration counts
        for x, y in zip(out[0], b):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("frac", [0.0, 0.003, 1.0])
def test_single_kernel_variant_at_extreme_plastic_fractions(ctx, oracle, frac):
    """Variant 2 feeds its Newton lanes from a per-wave queue filled by its own classification: no plastic point at all, a
    few per thousand (most tiles add nothing to the queue) and every point plastic (the queue is always full)."""
    pool_d, pool_s = mc_tracing_inputs(oracle, 20000, seed=12)
    ref = oracle.mohr_coulomb(pool_d, pool_s, nthreads=8)
    pl, el = np.flatnonzero(ref[3] > 0), np.flatnonzero(ref[3] <= 0)
    rng = np.random.default_rng(3)
    n = 40_000
    take = np.where(rng.random(n) < frac, rng.choice(pl, n), rng.choice(el, n))
    deps, sn = pool_d[take], pool_s[take]
    try:
        ctx.set_option("mc_variant", 2)
        got = run_device(ctx, deps, sn)
    finally:
        ctx.set_option("mc_variant", 2)
    mc_compare(got, tuple(r[take] for r in ref), f"single kernel, plastic fraction {frac}", sn)


def test_full_size_properties(ctx, oracle):
    """10^6 points (config 4 is 10^7; the properties are size independent): plastic points end on f = 0,
    elastic points return C_elas and the trial stress, and the tangent is the derivative of the stress map."""
    pool_d, pool_s = mc_tracing_inputs(oracle, 20000, seed=2)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 20000, 1_000_000)
    deps, sn = pool_d[idx] * rng.uniform(0.5, 1.0, (idx.size, 1)), pool_s[idx]
    Ct, s, it, y, nr, dl = run_device(ctx, deps, sn)
    Cel, _ = mc_elastic_matrices()
    el = y <= 0
    assert 0.2 < el.mean() < 0.9
    assert np.all(it[el] == 1) and np.array_equal(Ct[el], np.broadcast_to(Cel, Ct[el].shape))
    conv = ~el & (it < 200)
    assert conv.sum() > 0.99 * (~el).sum()
    f_after = oracle.mc_surface(s[conv][:50000])[0]
    assert np.max(np.abs(f_after)) < 1e-6
    assert np.all(dl[conv] > 0)
    sel = np.flatnonzero(conv)[:200]
    h = 1e-6
    v = rng.normal(size=(sel.size, 4))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    s2 = run_device(ctx, deps[sel] + h * v, sn[sel])[1]
    lin = h * np.einsum("nij,nj->ni", Ct[sel], v)
    err = np.linalg.norm(s2 - s[sel] - lin, axis=1) / np.linalg.norm(lin, axis=1)
    assert np.median(err) < 1e-3


def test_device_side_newton_summary(ctx, oracle):
    """dxo_mc_summary reproduces the reference's printed summary (:584-591) from device-resident diagnostics."""
    import torch

    n = 100_000
    pool_d, pool_s = mc_tracing_inputs(oracle, 20000, seed=4)
    rng = np.random.default_rng(4)
    idx = rng.integers(0, 20000, n)
    deps, sn = pool_d[idx], pool_s[idx]
    sn[7] = [0.1, 0.1, 0.1, 0.0]          # hydrostatic trial: f = NaN like the reference
    deps[7] = 0.0
    dev = torch.device("cuda:0")
    d, s0 = torch.from_numpy(deps).to(dev), torch.from_numpy(sn).to(dev)
    Ct = torch.empty(n * 16, dtype=torch.float64, device=dev)
    s = torch.empty(n * 4, dtype=torch.float64, device=dev)
    it = torch.empty(n, dtype=torch.int32, device=dev)
    y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.mohr_coulomb(params(), n, MEM_DEVICE, d.data_ptr(), s0.data_ptr(), Ct.data_ptr(), s.data_ptr(), it.data_ptr(),
                     y.data_ptr(), nr.data_ptr(), dl.data_ptr())
    summ = ctx.mc_summary(n, it, y, nr, nbins=201)
    it_h, y_h, nr_h = it.cpu().numpy(), y.cpu().numpy(), nr.cpu().numpy()
    u, c = np.unique(it_h, return_counts=True)
    assert np.array_equal(summ["unique_iters"], u) and np.array_equal(summ["counts"], c)
    assert summ["max_yielding"] == np.nanmax(y_h) and summ["max_norm_res"] == np.nanmax(nr_h)
    assert summ["nan_yielding"] == int(np.isnan(y_h).sum()) >= 1
    assert summ["nan_norm_res"] == int(np.isnan(nr_h).sum())
    with pytest.raises(ValueError, match="SIZE"):
        ctx.mc_summary(n, it, y, nr, nbins=5000)


def test_batches_larger_than_one_list_part(ctx, oracle):
    """The compacted-Newton schedule keeps int32 list entries, so batches beyond 2^30 points are processed in parts;
    `mc_part_points` shrinks the part size so that path runs here: same bits as the single-part run, including the
    ragged last part and parts without any plastic point."""
    deps, sn = mc_tracing_inputs(oracle, 30_011, seed=12, shear=0.1)
    deps[:5000] *= 1e-3                                      # a stretch that stays elastic: parts with an empty list
    whole = run_device(ctx, deps, sn)
    old = ctx.get_option("mc_part_points")
    try:
        for part in (4096, 64, 1000):                        # 1000 is rounded down to 960 (whole wave tiles)
            ctx.set_option("mc_part_points", part)
            split = run_device(ctx, deps, sn)
            for a, b in zip(whole, split):
                assert np.array_equal(a, b, equal_nan=True), part
    finally:
        ctx.set_option("mc_part_points", old)
