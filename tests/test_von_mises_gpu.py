"""Parity of the HIP von Mises kernels (through the C ABI) with the CPU oracle and the reference goldens."""
import numpy as np
import pytest

from conftest import assert_close_scaled, vm_inputs
from dolfinx_external_operator_amd import (
    MEM_DEVICE,
    MEM_HOST,
    Operand,
    QuadratureExternalOperator,
    VmParams,
    evaluate_external_operators,
    evaluate_operands,
    make_von_mises,
)

pytestmark = pytest.mark.gpu

# fp64. The HIP kernel uses the sparse structure of C_elas / deviatoric and FMA contraction, the oracle
# follows NumPy's dense mat-vecs: differences are rounding-level, measured against the array's scale.
RTOL = 1e-13

E, NU, SIGMA_0 = 70e3, 0.3, 250.0
H = E * (E / 100.0) / (E - E / 100.0)
PRM = VmParams(E, NU, SIGMA_0, H)


def run_host(ctx, deps, sigma_n, p, variant=1):
    n, d = deps.shape
    C = np.empty(n * d * d)
    s = np.empty(n * d)
    dp = np.empty(n)
    ctx.set_option("vm_variant", variant)
    ctx.von_mises(PRM, d, n, MEM_HOST, deps, sigma_n, p, C, s, dp)
    return C.reshape(n, d, d), s.reshape(n, d), dp


def run_device(ctx, deps, sigma_n, p, variant=1):
    import torch

    n, d = deps.shape
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (deps, sigma_n, p)]
    C = torch.empty(n * d * d, dtype=torch.float64, device=dev)
    s = torch.empty(n * d, dtype=torch.float64, device=dev)
    dp = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.set_option("vm_variant", variant)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, *(x.data_ptr() for x in t), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    return C.cpu().numpy().reshape(n, d, d), s.cpu().numpy().reshape(n, d), dp.cpu().numpy()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("name", ["von_mises_d4.npz", "von_mises_d6.npz"])
def test_reference_golden(ctx, golden, name, variant):
    g = golden(name)
    d = g["deps"].shape[-1]
    deps, sigma_n, p = g["deps"].reshape(-1, d), g["sigma_n"].reshape(-1, d), g["p"].reshape(-1)
    for runner in (run_host, run_device):
        C, s, dp = runner(ctx, deps, sigma_n, p, variant)
        assert_close_scaled(C, g["C_tang"], RTOL, f"C_tang {runner.__name__}")
        assert_close_scaled(s, g["sigma"], RTOL, f"sigma {runner.__name__}")
        assert_close_scaled(dp, g["dp"], RTOL, f"dp {runner.__name__}")
        assert np.isnan(C[4]).all() and np.isnan(s[4]).all()      # the reference's 0/0 point stays NaN
        elastic = (g["dp"].reshape(-1) == 0.0) & np.isfinite(s).all(axis=1)
        assert np.array_equal(C[elastic], np.broadcast_to(g["C_elas"], C[elastic].shape))  # exact C_elas


@pytest.mark.parametrize("d", [4, 6])
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 1000, 4097])
def test_ragged_sizes_against_oracle(ctx, oracle, d, n):
    deps, sigma_n, p = vm_inputs(n, d, seed=100 + n, plastic_scale=0.6)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    for variant in (0, 1):
        C, s, dp = run_device(ctx, deps, sigma_n, p, variant)
        if n == 0:
            assert C.size == 0 and s.size == 0 and dp.size == 0
            continue
        assert_close_scaled(C, Co, RTOL, "C_tang")
        assert_close_scaled(s, so, RTOL, "sigma")
        assert_close_scaled(dp, dpo, RTOL, "dp")


@pytest.mark.parametrize("d", [4, 6])
def test_outputs_do_not_overrun(ctx, d):
    """Guard words after each output must survive (partial last tile)."""
    import torch

    n = 130
    deps, sigma_n, p = vm_inputs(n, d, seed=3)
    dev = torch.device("cuda:0")
    t = [torch.from_numpy(a).to(dev) for a in (deps, sigma_n, p)]
    guard = 64
    outs = [torch.full((n * k + guard,), -7.0, dtype=torch.float64, device=dev) for k in (d * d, d, 1)]
    ctx.set_option("vm_variant", 1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, *(x.data_ptr() for x in t), *(o.data_ptr() for o in outs))
    torch.cuda.synchronize()
    for o, k in zip(outs, (d * d, d, 1)):
        assert torch.all(o[n * k:] == -7.0)
        assert not torch.any(o[: n * k] == -7.0)


@pytest.mark.parametrize("d", [4, 6])
def test_tiled_and_scalar_kernels_agree_at_scale(ctx, oracle, d):
    n = 200_000
    deps, sigma_n, p = vm_inputs(n, d, seed=5)
    C1, s1, dp1 = run_device(ctx, deps, sigma_n, p, 1)
    C0, s0, dp0 = run_device(ctx, deps, sigma_n, p, 0)
    assert_close_scaled(C1, C0, 1e-14, "C_tang v1 vs v0")
    assert_close_scaled(s1, s0, 1e-14, "sigma v1 vs v0")
    assert np.array_equal(dp1, dp0)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p, nthreads=8)
    assert_close_scaled(C1, Co, RTOL, "C_tang")
    assert_close_scaled(s1, so, RTOL, "sigma")
    assert_close_scaled(dp1, dpo, RTOL, "dp")


def test_unaligned_pointers_fall_back_to_scalar_kernel(ctx, oracle):
    import torch

    n, d = 777, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=9)
    dev = torch.device("cuda:0")
    big = torch.zeros(n * d + 1, dtype=torch.float64, device=dev)
    big[1:] = torch.from_numpy(deps.reshape(-1)).to(dev)   # 8-byte aligned, not 16
    sn = torch.from_numpy(sigma_n).to(dev)
    pp = torch.from_numpy(p).to(dev)
    C = torch.empty(n * d * d, dtype=torch.float64, device=dev)
    s = torch.empty(n * d, dtype=torch.float64, device=dev)
    dp = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.set_option("vm_variant", 1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, big.data_ptr() + 8, sn.data_ptr(), pp.data_ptr(), C.data_ptr(),
                  s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C.cpu().numpy(), Co, RTOL, "C_tang")
    with pytest.raises(ValueError, match="ALIGN"):
        ctx.von_mises(PRM, d, n, MEM_DEVICE, big.data_ptr() + 4, sn.data_ptr(), pp.data_ptr(), C.data_ptr(),
                      s.data_ptr(), dp.data_ptr())


def test_argument_validation(ctx):
    a = np.zeros(16)
    with pytest.raises(ValueError, match="DIM"):
        ctx.von_mises(PRM, 5, 1, MEM_HOST, a, a, a, a, a, a)
    with pytest.raises(ValueError, match="SIZE"):
        ctx.von_mises(PRM, 4, -1, MEM_HOST, a, a, a, a, a, a)
    with pytest.raises(ValueError, match="NULL"):
        ctx.von_mises(PRM, 4, 1, MEM_HOST, a, None, a, a, a, a)
    with pytest.raises(ValueError, match="MEM"):
        ctx.von_mises(PRM, 4, 1, 7, a, a, a, a, a, a)
    with pytest.raises(ValueError, match="OPTION"):
        ctx.set_option("no_such_knob", 1)


def test_host_pipeline_chunking(ctx, oracle):
    """Many small chunks through the 3-slot H2D/kernel/D2H ring must equal one big call."""
    n, d = 10_000, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=21)
    old = ctx.get_option("host_chunk_points")
    try:
        ctx.set_option("host_chunk_points", 640)
        C, s, dp = run_host(ctx, deps, sigma_n, p)
        t = ctx.last_timing()
        assert t["kernel_ms"] > 0 and t["h2d_ms"] > 0 and t["d2h_ms"] > 0 and t["total_ms"] > 0
    finally:
        ctx.set_option("host_chunk_points", old)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C, Co, RTOL, "C_tang")
    assert_close_scaled(s, so, RTOL, "sigma")
    assert_close_scaled(dp, dpo, RTOL, "dp")


def test_drop_in_external_function_through_evaluate_external_operators(ctx, oracle):
    """The demo's call sequence (demo_plasticity_von_mises.py:445-456) with the HIP-backed callback."""
    nc, nq, d = 50, 3, 4
    deps_full, sigma_n, p = vm_inputs(nc * nq, d, seed=33)
    deps_full = deps_full.reshape(nc, nq, d)
    state = {"sigma_n": sigma_n.reshape(-1).copy(), "p": p.copy()}
    sigma_external = make_von_mises(lambda: state["sigma_n"], lambda: state["p"], ctx=ctx)
    eps = Operand(lambda cells: deps_full[cells], "eps(Du)")
    sigma = QuadratureExternalOperator(eps, num_cells=nc, num_points=nq, value_shape=(d,),
                                       external_function=sigma_external)
    C_tang = QuadratureExternalOperator(eps, num_cells=nc, num_points=nq, value_shape=(d, d),
                                        external_function=sigma_external, derivatives=(1,))
    evaluated_operands = evaluate_operands([sigma])
    ((_, sigma_new, dp_new),) = evaluate_external_operators([C_tang], evaluated_operands)
    sigma.ref_coefficient.x.array[:] = sigma_new      # demo :453-454
    # the extras are views of the operator's pinned output buffers (reuse_outputs=True): copy before the next call,
    # exactly what the demo does with dp (:456)
    sigma_new, dp_new = sigma_new.copy(), dp_new.copy()
    Co, so, dpo = oracle.von_mises(deps_full, sigma_n, p)
    assert_close_scaled(C_tang.ref_coefficient.x.array, Co, RTOL, "C_tang coefficient")
    assert_close_scaled(sigma.ref_coefficient.x.array, so, RTOL, "sigma coefficient")
    assert_close_scaled(dp_new, dpo, RTOL, "dp")
    with pytest.raises(NotImplementedError, match="No external function is defined"):
        evaluate_external_operators([sigma], evaluated_operands)
    # the closure state is re-read at every call (load stepping mutates it, :564-565)
    state["p"] = p + dp_new
    state["sigma_n"] = sigma_new.copy()
    ((C2, s2, dp2),) = evaluate_external_operators([C_tang], evaluated_operands)
    Co2, so2, dpo2 = oracle.von_mises(deps_full, sigma_new.reshape(-1, d), p + dp_new)
    assert_close_scaled(s2, so2, RTOL, "sigma step 2")
    assert_close_scaled(dp2, dpo2, RTOL, "dp step 2")


def test_device_resident_tensors(ctx, oracle):
    import torch

    nc, nq, d = 1000, 8, 6
    deps, sigma_n, p = vm_inputs(nc * nq, d, seed=44)
    dev = torch.device("cuda:0")
    sn_t = torch.from_numpy(sigma_n).to(dev)
    p_t = torch.from_numpy(p).to(dev)
    ext = make_von_mises(sn_t, p_t, ctx=ctx)
    C, s, dp = ext((1,))(torch.from_numpy(deps.reshape(nc, nq, d)).to(dev))
    assert C.is_cuda and C.shape == (nc * nq * d * d,)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C.cpu().numpy(), Co, RTOL, "C_tang")
    assert_close_scaled(s.cpu().numpy(), so, RTOL, "sigma")


@pytest.mark.parametrize("d", [6])
def test_full_size_properties(ctx, d):
    """BASELINE config 2 size (10^6 points, d = 6): properties that need no oracle.
    plastic points land on the yield surface, elastic points return C_elas exactly, tangent symmetric,
    trace(sigma) is the elastic trace (the return is purely deviatoric)."""
    import torch

    n = 1_000_000
    deps, sigma_n, p = vm_inputs(n, d, seed=0, plastic_scale=0.7)
    C, s, dp = run_device(ctx, deps, sigma_n, p, 1)
    dev = s.copy()
    dev[:, :3] -= s[:, :3].mean(axis=1, keepdims=True)
    seq = np.sqrt(1.5 * np.sum(dev * dev, axis=1))
    plastic = dp > 0
    assert 0.05 < plastic.mean() < 0.999
    f = seq - SIGMA_0 - H * (p + dp)
    assert np.max(np.abs(f[plastic])) < 1e-9 * SIGMA_0
    assert np.all(f[~plastic] <= 1e-9)
    lmbda = E * NU / (1 + NU) / (1 - 2 * NU)
    mu = E / 2 / (1 + NU)
    C_el = np.zeros((d, d))
    C_el[:3, :3] = lmbda
    C_el[np.arange(d), np.arange(d)] += 2 * mu
    assert np.array_equal(C[~plastic], np.broadcast_to(C_el, C[~plastic].shape))
    assert np.max(np.abs(C - np.transpose(C, (0, 2, 1)))) < 1e-10 * E
    tr_el = sigma_n[:, :3].sum(1) + (3 * lmbda + 2 * mu) * deps[:, :3].sum(1)
    assert np.max(np.abs(s[:, :3].sum(1) - tr_el)) < 1e-10 * np.max(np.abs(tr_el))
    # consistent tangent: finite-difference check of d sigma / d eps on a handful of plastic points
    idx = np.flatnonzero(plastic)[:64]
    h = 1e-7
    for k in range(d):
        e2 = deps[idx].copy()
        e2[:, k] += h
        _, s2, _ = run_device(ctx, e2, sigma_n[idx], p[idx], 1)
        fd = (s2 - s[idx]) / h
        assert np.max(np.abs(fd - C[idx][:, :, k])) < 2e-5 * E


@pytest.mark.parametrize("d", [4, 6])
def test_device_resident_load_history(ctx, oracle, d):
    """Five load steps with sigma_n, p resident on the device (dxo_device_alloc / dxo_copy, no torch) and the fused
    history update dxo_vm_commit_state, against the oracle driven by the reference's two NumPy statements
    (demo_plasticity_von_mises.py:564-565). Checks the state after every step, not only the last."""
    n = 20_011
    deps0, sigma_n, p = vm_inputs(n, d, seed=77)
    sigma_n *= 0.0
    p *= 0.0
    nb = {"deps": n * d * 8, "sigma_n": n * d * 8, "p": n * 8, "C": n * d * d * 8, "sigma": n * d * 8, "dp": n * 8}
    dev = {k: ctx.device_alloc(v) for k, v in nb.items()}
    try:
        ctx.copy(dev["sigma_n"], sigma_n, nb["sigma_n"], 0)
        ctx.copy(dev["p"], p, nb["p"], 0)
        s_ref, p_ref = sigma_n.copy(), p.copy()
        got_s, got_p = np.empty_like(sigma_n), np.empty_like(p)
        C = np.empty((n, d, d))
        for step in range(5):
            deps = np.ascontiguousarray(deps0 * (0.4 + 0.3 * step) * (-1.0 if step == 3 else 1.0))   # unloading at step 3
            ctx.copy(dev["deps"], deps, nb["deps"], 0)
            ctx.von_mises(PRM, d, n, MEM_DEVICE, dev["deps"], dev["sigma_n"], dev["p"], dev["C"], dev["sigma"], dev["dp"])
            ctx.vm_commit_state(d, n, dev["p"], dev["dp"], dev["sigma_n"], dev["sigma"])
            ctx.synchronize()
            C_o, s_o, dp_o = oracle.von_mises(deps, s_ref, p_ref)
            p_ref += dp_o.reshape(-1)                    # :564
            s_ref[:] = s_o.reshape(n, d)                 # :565
            ctx.copy(got_s, dev["sigma_n"], nb["sigma_n"], 1)
            ctx.copy(got_p, dev["p"], nb["p"], 1)
            ctx.copy(C, dev["C"], nb["C"], 1)
            assert_close_scaled(got_s, s_ref, RTOL, f"sigma_n after step {step}")
            assert_close_scaled(got_p, p_ref, RTOL, f"p after step {step}")
            assert_close_scaled(C, C_o, RTOL, f"C_tang at step {step}")
        assert (p_ref > 0).mean() > 0.5                  # the history really went plastic
    finally:
        for a in dev.values():
            ctx.device_free(a)


def test_commit_state_torch_and_errors(ctx):
    import torch

    from dolfinx_external_operator_amd import von_mises_commit_state

    n, d = 1000, 6
    g = torch.Generator().manual_seed(1)
    p, dp = (torch.rand(n, dtype=torch.float64, generator=g).cuda() for _ in range(2))
    sn, s = (torch.rand(n * d, dtype=torch.float64, generator=g).cuda() for _ in range(2))
    want_p = (p + dp).cpu()
    von_mises_commit_state(p, dp, sn, s, ctx=ctx)
    torch.cuda.synchronize()
    assert torch.equal(p.cpu(), want_p) and torch.equal(sn, s)
    with pytest.raises(ValueError):
        von_mises_commit_state(p, dp[:-1].contiguous(), sn, s, ctx=ctx)
    with pytest.raises(TypeError):
        von_mises_commit_state(p.cpu(), dp, sn, s, ctx=ctx)
    with pytest.raises(ValueError):           # argument errors (rc < 0) map to ValueError, HIP errors to DxoError
        ctx.vm_commit_state(5, n, p.data_ptr(), dp.data_ptr(), sn.data_ptr(), s.data_ptr())
    with pytest.raises(ValueError):
        ctx.copy(p.data_ptr(), dp.data_ptr(), 8, 7)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("d,n", [(4, 1), (4, 4097), (6, 63), (6, 50_001)])
def test_expand_tangent_from_returned_state(ctx, oracle, d, n, variant):
    """dxo_vm_expand_tangent(sigma, dp) rebuilds what dxo_von_mises wrote as C_tang: to rounding on plastic points,
    bit for bit on elastic ones (C_elas), same NaN pattern; and it meets the oracle's tangent at the usual 1e-13.
    This is the property the compact multi-GPU gather rests on (sharding.gather_von_mises_compact)."""
    deps, sigma_n, p = vm_inputs(n, d, seed=5 + n)
    deps[: n // 3] *= 0.2
    sigma_n[: n // 3] *= 0.2
    if n > 10:
        deps[7], sigma_n[7] = 0.0, 0.0             # zero trial stress: sigma_eq = 0 -> the reference's 0/0 (:318-319)
    C_k, s_k, dp_k = run_device(ctx, deps, sigma_n, p, variant)
    ctx.set_option("vm_variant", variant)
    C_x = np.empty((n, d, d))
    ctx.vm_expand_tangent(PRM, d, n, MEM_HOST, np.ascontiguousarray(s_k), np.ascontiguousarray(dp_k), C_x)
    assert_close_scaled(C_x, C_k, 1e-14, "expand vs kernel")
    el = (dp_k == 0) & ~np.isnan(s_k).any(axis=1)
    assert el.any() or n < 3
    assert np.array_equal(C_x[el], C_k[el])
    if n > 10:
        assert np.isnan(C_x[7]).all() and np.isnan(C_k[7]).all()
    with np.errstate(all="ignore"):
        C_o, _, _ = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C_x, C_o, RTOL, "expand vs oracle")


def test_expand_tangent_device_pointers_and_errors(ctx):
    import torch

    n, d = 100_000, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=2)
    C_k, s_k, dp_k = run_device(ctx, deps, sigma_n, p)
    s_t, dp_t = torch.from_numpy(s_k).cuda(), torch.from_numpy(dp_k).cuda()
    C_t = torch.empty(n * d * d, dtype=torch.float64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.vm_expand_tangent(PRM, d, n, MEM_DEVICE, s_t.data_ptr(), dp_t.data_ptr(), C_t.data_ptr())
    torch.cuda.synchronize()
    assert_close_scaled(C_t.cpu().numpy(), C_k, 1e-14, "expand (device)")
    # 8-byte-aligned but not 16-byte-aligned views take the scalar kernel
    s_o = torch.empty(n * d + 1, dtype=torch.float64, device="cuda")
    s_o[1:] = s_t.reshape(-1)
    ctx.vm_expand_tangent(PRM, d, n, MEM_DEVICE, s_o.data_ptr() + 8, dp_t.data_ptr(), C_t.data_ptr())
    torch.cuda.synchronize()
    assert_close_scaled(C_t.cpu().numpy(), C_k, 1e-14, "expand (unaligned)")
    with pytest.raises(ValueError):
        ctx.vm_expand_tangent(PRM, 5, n, MEM_DEVICE, s_t.data_ptr(), dp_t.data_ptr(), C_t.data_ptr())
    with pytest.raises(ValueError):
        ctx.vm_expand_tangent(PRM, d, n, MEM_DEVICE, None, dp_t.data_ptr(), C_t.data_ptr())
    ctx.vm_expand_tangent(PRM, d, 0, MEM_DEVICE, None, None, None)


def test_reference_load_history_golden(ctx, golden):
    """The reference's own six-step load history (tests/golden/von_mises_history_d4.npz): the drop-in callback with the
    state update done as in the demo (:564-565), and the device-resident variant with dxo_vm_commit_state, both
    carried on their own accumulated state."""
    import torch

    g = golden("von_mises_history_d4.npz")
    n_steps, d = int(g["n_steps"]), 4
    n = g["dp_0"].size
    sigma_n, p = np.zeros(n * d), np.zeros(n)                     # the closure state the callback re-reads (:347-348)
    fn = make_von_mises(lambda: sigma_n, lambda: p, ctx=ctx)
    dev = [torch.zeros(n * d, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")]
    out = [torch.empty(n * d * d, dtype=torch.float64, device="cuda"), torch.empty(n * d, dtype=torch.float64, device="cuda"),
           torch.empty(n, dtype=torch.float64, device="cuda")]
    for k in range(n_steps):
        deps = g[f"deps_{k}"]
        C, s, dp = fn((1,))(deps)
        assert_close_scaled(C, g[f"C_tang_{k}"], RTOL, f"C_tang step {k}")
        assert_close_scaled(s, g[f"sigma_{k}"], RTOL, f"sigma step {k}")
        assert_close_scaled(dp, g[f"dp_{k}"], RTOL, f"dp step {k}")
        p += dp                                                    # :564
        sigma_n[:] = s                                             # :565
        e = torch.from_numpy(np.ascontiguousarray(deps).reshape(-1)).cuda()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.von_mises(PRM, d, n, MEM_DEVICE, e.data_ptr(), dev[0].data_ptr(), dev[1].data_ptr(), out[0].data_ptr(),
                      out[1].data_ptr(), out[2].data_ptr())
        ctx.vm_commit_state(d, n, dev[1].data_ptr(), out[2].data_ptr(), dev[0].data_ptr(), out[1].data_ptr())
        torch.cuda.synchronize()
        assert_close_scaled(dev[1].cpu().numpy(), g[f"p_after_{k}"], RTOL, f"device p after step {k}")
        assert_close_scaled(dev[0].cpu().numpy(), g[f"sigma_n_after_{k}"], RTOL, f"device sigma_n after step {k}")


def test_two_contexts_from_two_threads(oracle):
    """SURVEY.md 8b threading row: entry points are blocking and re-entrant per ctx; ctypes releases the GIL, so two
    Python threads with their own contexts really run concurrently. Each checks its own results."""
    import threading

    from dolfinx_external_operator_amd import Context

    errors = []

    def worker(seed, d):
        try:
            c = Context(0)
            try:
                deps, sigma_n, p = vm_inputs(200_000, d, seed=seed)
                Co, so, dpo = oracle.von_mises(deps, sigma_n, p, nthreads=2)
                for _ in range(4):
                    C, s, dp = run_host(c, deps, sigma_n, p)
                    assert_close_scaled(C, Co, RTOL, "C_tang")
                    assert_close_scaled(s, so, RTOL, "sigma")
                    assert_close_scaled(dp, dpo, RTOL, "dp")
            finally:
                c.close()
        except Exception as exc:   # noqa: BLE001 - reported to the main thread
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(31, 4)), threading.Thread(target=worker, args=(32, 6))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors


@pytest.mark.parametrize("n", [1, 600, 6144])
def test_small_batch_path_equals_chunked_pipeline(ctx, oracle, n):
    """Host batches below host_small_bytes take the single-copy staging path; it must return exactly what the chunked
    pipeline returns (same kernel, same data), with guard words intact, for every kernel that uses the pipeline."""
    from dolfinx_external_operator_amd import MEM_HOST as HOST

    deps, sigma_n, p = vm_inputs(n, 4, seed=n)
    old = ctx.get_option("host_small_bytes")
    res = {}
    try:
        for small in (0, 1 << 20):
            ctx.set_option("host_small_bytes", small)
            C, s, dp = np.full(n * 16 + 2, -5.0), np.full(n * 4 + 2, -5.0), np.full(n + 2, -5.0)
            ctx.von_mises(PRM, 4, n, HOST, deps, sigma_n, p, C, s, dp)
            assert C[-1] == -5.0 and s[-2] == -5.0 and dp[-1] == -5.0
            res[small] = (C.copy(), s.copy(), dp.copy())
            T = np.random.default_rng(n).random(n) + 0.5
            sg = np.random.default_rng(n + 1).normal(size=(n, 2))
            q, dT, ds = np.empty(n * 2), np.empty(n * 2), np.empty(n * 4)
            ctx.heat(1.0, 1.0, 2, n, HOST, T, sg, q, dT, ds)
            res[("heat", small)] = (q, dT, ds)
        ctx.set_option("timing", 1)
        ctx.von_mises(PRM, 4, n, HOST, deps, sigma_n, p, C, s, dp)
        t = ctx.last_timing()
        assert t["total_ms"] > 0 and t["kernel_ms"] > 0
    finally:
        ctx.set_option("timing", 0)
        ctx.set_option("host_small_bytes", old)
    for a, b in zip(res[0], res[1 << 20]):
        assert np.array_equal(a, b, equal_nan=True)
    for a, b in zip(res[("heat", 0)], res[("heat", 1 << 20)]):
        assert np.array_equal(a, b)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(res[1 << 20][0][: n * 16], Co, RTOL, "C_tang small path")


@pytest.mark.parametrize("E_,nu_,s0_,H_", [(210e3, 0.25, 400.0, 2000.0), (10.0, 0.45, 0.05, 0.0), (70e3, 0.0, 250.0, 70e3)])
def test_other_material_parameters(ctx, oracle, E_, nu_, s0_, H_):
    """Nothing in the kernels is specialised to the demo's constants (:185-188): steel-like values, a soft nearly
    incompressible material with perfect plasticity (H = 0), and nu = 0 with very stiff hardening."""
    prm = VmParams(E_, nu_, s0_, H_)
    for d in (4, 6):
        n = 5000
        rng = np.random.Generator(np.random.PCG64(d))
        deps = rng.normal(0.0, 0.35 * s0_ / E_, size=(n, d))
        sigma_n = rng.normal(0.0, 0.25 * s0_, size=(n, d))
        p = np.abs(rng.normal(0.0, 1e-3, size=n))
        Co, so, dpo = oracle.von_mises(deps, sigma_n, p, E=E_, nu=nu_, sigma_0=s0_, H=H_)
        assert 0.1 < (dpo > 0).mean() < 0.99
        C, s, dp = np.empty(n * d * d), np.empty(n * d), np.empty(n)
        ctx.von_mises(prm, d, n, MEM_HOST, deps, sigma_n, p, C, s, dp)
        assert_close_scaled(C, Co, RTOL, "C_tang")
        assert_close_scaled(s, so, RTOL, "sigma")
        assert_close_scaled(dp, dpo, RTOL, "dp")
        Cx = np.empty(n * d * d)
        ctx.vm_expand_tangent(prm, d, n, MEM_HOST, s, dp, Cx)
        assert_close_scaled(Cx, Co, RTOL, "tangent from state")


@pytest.mark.parametrize("n, d", [(10_001, 6), (4_097, 4), (63, 6), (30_000, 4)])
def test_small_host_path_in_place_pieces_and_staging_give_the_device_result(ctx, n, d):
    """Round 6 (csrc/dxo_ctx.hip, small path of the host pipeline): batches up to 8 MiB run as kernels on page-locked host memory.
    Every mixture must leave the bits a device call leaves: (a) pageable arrays through the pinned staging block, one piece; (b) four
    pieces on ragged borders; (c) output arrays that are page-locked blocks of the library (written in place) with pageable inputs;
    (d) everything page-locked — inputs registered by the caller (Context.pin), outputs from the pool — nothing is copied at all;
    (e) the DMA form (host_zero_copy_bytes = 0: one packed H2D + one packed D2H). Guard words behind every output."""
    deps, sigma_n, p = vm_inputs(n, d, seed=31)
    Cd, sd, dpd = run_device(ctx, deps, sigma_n, p)
    keys = ("host_zero_copy_bytes", "host_zero_copy_piece_bytes", "host_small_bytes")
    old = {k: ctx.get_option(k) for k in keys}
    sizes = (n * d * d, n * d, n)

    def check(outs, what):
        for got, ref, m in zip(outs, (Cd, sd, dpd), sizes):
            assert np.array_equal(got[:m], np.asarray(ref).reshape(-1), equal_nan=True), what
            assert np.all(got[m:] == -7.0), what + ": guard words"

    def pageable():
        return [np.full(m + 8, -7.0) for m in sizes]

    try:
        for label, opts in (("one piece", {"host_zero_copy_piece_bytes": 0}), ("four pieces", {"host_zero_copy_piece_bytes": 4096}),
                            ("DMA form", {"host_zero_copy_bytes": 0})):
            for k, v in {**old, **opts}.items():
                ctx.set_option(k, v)
            outs = pageable()
            ctx.von_mises(PRM, d, n, MEM_HOST, deps, sigma_n, p, *outs)
            check(outs, label + ", pageable arrays")
            pool = [ctx.pinned_recycled(m + 8) for m in sizes]          # what the factories hand out: written in place
            for a in pool:
                a[:] = -7.0
            ctx.von_mises(PRM, d, n, MEM_HOST, deps, sigma_n, p, *pool)
            check(pool, label + ", page-locked outputs")
        for k, v in {**old, "host_zero_copy_piece_bytes": 4096}.items():
            ctx.set_option(k, v)
        ins = [np.ascontiguousarray(a).copy() for a in (deps, sigma_n, p)]
        for a in ins:
            ctx.pin(a)
        try:
            pool = [ctx.pinned_recycled(m + 8) for m in sizes]
            for a in pool:
                a[:] = -7.0
            ctx.von_mises(PRM, d, n, MEM_HOST, *ins, *pool)
            check(pool, "everything page-locked")
            assert all(np.array_equal(a, b) for a, b in zip(ins, (deps, sigma_n, p)))      # inputs read in place, untouched
        finally:
            for a in ins:
                ctx.unpin(a)
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
