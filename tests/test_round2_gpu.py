"""Round-2 additions on the GPU: output arena, host tangent rebuild, fresh-output semantics, zero-copy (device tensor)
paths of every factory, dofmap validation in dxo_assign, serialised entry points."""
import threading

import numpy as np
import pytest

from conftest import assert_close_scaled, mc_compare, mc_tracing_inputs, vm_inputs
from dolfinx_external_operator_amd import (
    MEM_DEVICE,
    MEM_HOST,
    AssignDesc,
    VmParams,
    make_heat,
    make_icnn,
    make_isihara,
    make_mohr_coulomb,
    make_von_mises,
)

pytestmark = pytest.mark.gpu

E, NU, SIGMA_0 = 70e3, 0.3, 250.0
H = E * (E / 100.0) / (E - E / 100.0)
PRM = VmParams(E, NU, SIGMA_0, H)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


# ------------------------------------------------------------------------------------------------- output arena
def test_output_arena_small_block_is_plain_and_usable(ctx):
    import torch

    C, s, dp = ctx.output_tensors((4096 * 36, 4096 * 6, 4096))
    info = C.dxo_block.info
    assert info["mode"] == "hipMalloc" and info["chosen"] == -1            # below placement_min_bytes: no calibration
    assert C.is_cuda and C.dtype == torch.float64 and C.numel() == 4096 * 36
    assert C.data_ptr() % 256 == 0 and s.data_ptr() % 256 == 0 and dp.data_ptr() % 256 == 0
    assert s.data_ptr() >= C.data_ptr() + C.numel() * 8 and dp.data_ptr() >= s.data_ptr() + s.numel() * 8
    C.fill_(1.5)
    s.fill_(2.5)
    assert float(C.sum()) == 1.5 * C.numel() and float(s[-1]) == 2.5


@pytest.mark.parametrize("vmm", [1, 0])
def test_output_arena_calibrates_and_kernel_results_are_unchanged(ctx, oracle, vmm, mode=2):
    """A 1.1 GB block goes through the calibration (candidates side by side — hipMalloc blocks and, with placement_vmm,
    virtual ranges backed by 2 MB physical chunks — the fastest kept); the kernel writing into it gives the oracle's
    numbers; the record names the chosen candidate and its kind."""
    import torch

    n, d = 3_200_000, 6          # 344 B/point -> 1.10 GB
    old = {k: ctx.get_option(k) for k in ("placement_mode", "placement_candidates", "placement_vmm")}
    ctx.set_option("placement_mode", mode)
    ctx.set_option("placement_candidates", 4)
    ctx.set_option("placement_vmm", vmm)
    try:
        C, s, dp = ctx.output_tensors((n * d * d, n * d, n))
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
    info = C.dxo_block.info
    assert info["mode"] == "candidates", info
    want = (["hipMalloc", "2MB_chunks", "2MB_chunks", "2MB_chunks"] * 2)[: info["candidates"]] if vmm else ["hipMalloc"] * info["candidates"]
    assert info["kinds"][:4] == want[:4] and info["chosen_kind"] == info["kinds"][info["chosen"]], info
    assert 1 <= info["candidates"] <= 5 and 0 <= info["chosen"] < info["candidates"]   # 4 + the late lone allocation
    assert all(b > 1000.0 for b in info["probe_GBps"]), info             # every candidate was really timed (GB/s)
    assert info["chosen_GBps"] > 1000.0                                   # the final round's rate of the block kept
    g = torch.Generator(device="cuda:0").manual_seed(5)
    deps = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 3e-3, generator=g)
    sigma_n = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 100.0, generator=g)
    p = torch.empty(n, dtype=torch.float64, device="cuda:0").normal_(0, 1e-3, generator=g).abs_()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    idx = torch.arange(0, n, 997, device="cuda:0")                          # strided sample incl. point 0
    idx = torch.cat([idx, torch.tensor([n - 1], device="cuda:0")])
    Co, so, dpo = oracle.von_mises(deps[idx].cpu().numpy(), sigma_n[idx].cpu().numpy(), p[idx].cpu().numpy())
    assert_close_scaled(C.view(n, d * d)[idx].cpu().numpy(), Co, 1e-13, "C_tang in the arena")
    assert_close_scaled(s.view(n, d)[idx].cpu().numpy(), so, 1e-13, "sigma in the arena")
    assert_close_scaled(dp[idx].cpu().numpy(), dpo, 1e-13, "dp in the arena")
    ptr = C.dxo_block.ptr
    del C, s, dp
    import gc

    gc.collect()
    with pytest.raises(ValueError):       # the block was returned to the library when its last tensor went away
        ctx.output_info(ptr)


def test_factory_arena_outputs(ctx, oracle):
    import torch

    nc, nq, d = 500, 8, 6
    deps, sigma_n, p = vm_inputs(nc * nq, d, seed=71)
    ext = make_von_mises(_dev(sigma_n), _dev(p), ctx=ctx)
    out = ext.arena(nc * nq, d)
    C, s, dp = ext((1,))(_dev(deps.reshape(nc, nq, d)), out=out)
    assert C.data_ptr() == out[0].data_ptr() and dp.data_ptr() == out[2].data_ptr()
    torch.cuda.synchronize()
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C.cpu().numpy(), Co, 1e-13, "C_tang")
    assert_close_scaled(dp.cpu().numpy(), dpo, 1e-13, "dp")


# ------------------------------------------------------------------------------------------------- host tangent rebuild
@pytest.mark.parametrize("d", [4, 6])
@pytest.mark.parametrize("n", [1, 63, 1000, 300_001])
def test_host_tangent_rebuild_matches_the_device_tangent(ctx, d, n):
    """vm_host_tangent = 1: only (sigma, dp) cross PCIe and C_tang is rebuilt by the host threads. sigma and dp are
    bit-identical to the copy mode; the tangent agrees to rounding (1e-13 of its scale), elastic points exactly."""
    deps, sigma_n, p = vm_inputs(n, d, seed=100 + n % 97 + d)
    outs = {}
    old_chunk, old_min = ctx.get_option("host_chunk_points"), ctx.get_option("vm_rebuild_min_points")
    ctx.set_option("host_chunk_points", 65536)     # several chunks at the largest size: the hook runs per chunk
    ctx.set_option("vm_rebuild_min_points", 0)     # also the small sizes go through the rebuild (default: >= 2^16 points)
    try:
        for mode in (0, 1):
            ctx.set_option("vm_host_tangent", mode)
            C, s, dp = np.full(n * d * d + 8, -7.0), np.empty(n * d), np.empty(n)
            ctx.von_mises(PRM, d, n, MEM_HOST, deps, sigma_n, p, C, s, dp)
            assert np.all(C[n * d * d:] == -7.0)                              # guard words behind the array
            outs[mode] = (C[: n * d * d].copy(), s, dp)
    finally:
        ctx.set_option("vm_host_tangent", 0)
        ctx.set_option("host_chunk_points", old_chunk)
        ctx.set_option("vm_rebuild_min_points", old_min)
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert_close_scaled(outs[1][0], outs[0][0], 1e-13, "host-rebuilt C_tang")
    el = outs[0][2] == 0.0
    C0, C1 = outs[0][0].reshape(n, -1), outs[1][0].reshape(n, -1)
    assert np.array_equal(C0[el], C1[el])                                      # elastic points: C_elas, bit for bit


def test_host_tangent_rebuild_through_the_factory(ctx, oracle):
    nc, nq, d = 400, 8, 6
    deps, sigma_n, p = vm_inputs(nc * nq, d, seed=9)
    ext = make_von_mises(sigma_n, p, ctx=ctx, host_tangent="rebuild")
    old_min = ctx.get_option("vm_rebuild_min_points")
    ctx.set_option("vm_rebuild_min_points", 0)
    try:
        C, s, dp = ext((1,))(deps.reshape(nc, nq, d))
    finally:
        ctx.set_option("vm_rebuild_min_points", old_min)
    assert ctx.get_option("vm_host_tangent") == 0                              # the option does not leak out of the call
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(C, Co, 1e-13, "C_tang")
    assert_close_scaled(s, so, 1e-13, "sigma")
    assert_close_scaled(dp, dpo, 1e-13, "dp")


# ------------------------------------------------------------------------------------------------- fresh outputs
def test_results_of_successive_calls_do_not_alias(ctx):
    """Reference semantics (demo_plasticity_von_mises.py:352: fresh arrays): a = f(x0); b = f(x1); b - a is the
    difference, not zero. Memory is recycled only after the caller has dropped every view."""
    nc, nq, d = 100, 4, 4
    deps, sigma_n, p = vm_inputs(nc * nq, d, seed=3)
    ext = make_von_mises(sigma_n, p, ctx=ctx)
    f = ext((1,))
    a = f(deps.reshape(nc, nq, d))
    keep = a[1].copy()
    b = f(2.0 * deps.reshape(nc, nq, d))
    assert a[1].ctypes.data != b[1].ctypes.data
    assert np.array_equal(a[1], keep) and not np.array_equal(a[1], b[1])
    addr = b[1].ctypes.data
    view = b[1].reshape(-1, d)[5:7]            # a view keeps the block alive
    del a, b
    c1 = f(deps.reshape(nc, nq, d))
    assert addr not in (x.ctypes.data for x in c1)
    del view, c1
    import gc

    gc.collect()
    c2 = f(deps.reshape(nc, nq, d))            # now the pool may hand the same page-locked blocks out again
    assert np.array_equal(c2[1], keep)


def test_reuse_outputs_opt_in_never_frees_under_a_view(ctx):
    d = 4
    deps, sigma_n, p = vm_inputs(256, d, seed=4)
    state = {"s": sigma_n, "p": p}
    ext = make_von_mises(lambda: state["s"], lambda: state["p"], ctx=ctx, reuse_outputs=True)
    a = ext((1,))(deps.reshape(32, 8, d))
    keep = a[0].copy()
    b = ext((1,))(deps.reshape(32, 8, d))
    assert a[0].ctypes.data == b[0].ctypes.data                # opt-in: the same buffers, overwritten
    deps2, state["s"], state["p"] = vm_inputs(512, d, seed=5)
    c = ext((1,))(deps2.reshape(64, 8, d))                      # batch size changed: old buffers are retired, not freed
    assert np.array_equal(a[0], keep) and c[0].size == 512 * d * d


# ------------------------------------------------------------------------------------------------- zero-copy factories
def test_heat_device_tensors(ctx, golden):
    import torch

    g = golden("heat_c1.npz")
    T, sig = g["T"], g["sigma"]
    q_ext = make_heat(ctx=ctx)
    for multi, key in (((0, 0), "q"), ((1, 0), "dqdT"), ((0, 1), "dqdsigma")):
        out = q_ext(multi)(_dev(T), _dev(sig))
        assert out.is_cuda and out.dtype == torch.float64
        assert_close_scaled(out.cpu().numpy(), g[key], 1e-15, key)
    with pytest.raises(TypeError):
        q_ext((0, 0))(_dev(T), sig)        # mixed host / device operands


def test_mohr_coulomb_device_tensors(ctx, oracle):
    n = 6000
    deps, sigma_n = mc_tracing_inputs(oracle, n, seed=21)
    seen = {}
    ext = make_mohr_coulomb(_dev(sigma_n), ctx=ctx, on_summary=seen.update)
    C, s = ext((1,))(_dev(deps))
    niter, yielding, norm_res, dlambda = ext.last_state
    assert C.is_cuda and niter.is_cuda and niter.dtype.is_floating_point is False
    ref = oracle.mohr_coulomb(deps, sigma_n, nthreads=8)
    got = (C.cpu().numpy().reshape(n, 4, 4), s.cpu().numpy().reshape(n, 4), niter.cpu().numpy(), yielding.cpu().numpy(),
           norm_res.cpu().numpy(), dlambda.cpu().numpy())
    mc_compare(got, ref, "device tensors", sigma_n)
    it_u, it_c = np.unique(ref[2], return_counts=True)          # the summary the reference prints (:584-591), reduced on the GPU
    assert np.array_equal(seen["unique_iters"], it_u) and np.array_equal(seen["counts"], it_c)
    assert abs(seen["max_yielding"] - np.max(ref[3])) <= 1e-12 * max(1.0, abs(np.max(ref[3])))
    # without diagnostics nothing but (C_tang, sigma) is allocated
    ext2 = make_mohr_coulomb(_dev(sigma_n), ctx=ctx, diagnostics=False)
    C2, s2 = ext2((1,))(_dev(deps))
    assert ext2.last_state == (None, None, None, None)
    assert np.array_equal(C2.cpu().numpy(), C.cpu().numpy())


def test_icnn_and_isihara_device_tensors(ctx, golden):
    g = golden("icnn_isihara.npz")
    w = golden("icnn_isihara_weights.npz")
    F = np.asarray(g["F"], dtype=np.float64).reshape(-1, 2, 2)
    ext = make_icnn({k: w[k] for k in w.files}, ctx=ctx)
    dP_h, P_h = ext((1,))(F)
    dP_d, P_d = ext((1,))(_dev(F))
    assert dP_d.is_cuda
    assert np.array_equal(dP_d.cpu().numpy(), dP_h) and np.array_equal(P_d.cpu().numpy(), P_h)   # same kernel, same bits
    isi = make_isihara(ctx=ctx)
    a_h = isi((1,))(F)
    a_d = isi((1,))(_dev(F))
    assert np.array_equal(a_d[0].cpu().numpy(), a_h[0]) and np.array_equal(a_d[1].cpu().numpy(), a_h[1])


# ------------------------------------------------------------------------------------------------- dxo_assign validation
def test_assign_rejects_out_of_range_dofs(ctx):
    import torch

    n_cells, n_pts, vs, coeff_size = 40, 3, 2, 100
    rng = np.random.default_rng(0)
    dofs = rng.integers(0, coeff_size, n_cells * n_pts * vs).astype(np.int32)
    vals = rng.normal(size=n_cells * n_pts * vs)
    desc = AssignDesc(n_cells, n_pts, vs, 0, n_pts, vs, 0)
    coeff = torch.zeros(coeff_size + 16, dtype=torch.float64, device="cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_dofs, d_vals = _dev(dofs), _dev(vals)       # keep the tensors alive: only raw pointers cross the C ABI
    ctx.assign(desc, d_dofs.data_ptr(), d_vals.data_ptr(), coeff.data_ptr(), coeff_size)   # in range: fine
    bad = dofs.copy()
    bad[7], bad[100] = coeff_size + 3, -2
    d_bad = _dev(bad)
    coeff2 = torch.zeros(coeff_size + 16, dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="2 flat_dofs entries outside"):
        ctx.assign(desc, d_bad.data_ptr(), d_vals.data_ptr(), coeff2.data_ptr(), coeff_size)
    torch.cuda.synchronize()
    assert float(coeff2[coeff_size:].abs().sum()) == 0.0        # nothing was written past the coefficient
    expect = np.zeros(coeff_size)
    ok = (bad >= 0) & (bad < coeff_size)
    expect[bad[ok]] = vals[ok]                                   # NumPy: last writer wins
    assert np.array_equal(coeff2[:coeff_size].cpu().numpy(), expect)


# ------------------------------------------------------------------------------------------------- serialised entry points
def test_two_threads_on_one_context(ctx, oracle):
    """ctypes releases the GIL; every C entry point takes the context's mutex, so two Python threads sharing the
    default context get correct (serialised) results instead of racing on slot buffers and scratch."""
    d = 4
    jobs = []
    for seed in (1, 2):
        deps, sigma_n, p = vm_inputs(40_000, d, seed=seed)
        jobs.append((deps, sigma_n, p, make_von_mises(sigma_n, p, ctx=ctx)))
    res = [None, None]

    def work(k):
        for _ in range(5):
            res[k] = jobs[k][3]((1,))(jobs[k][0].reshape(-1, 8, d))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(2):
        Co, so, dpo = oracle.von_mises(*jobs[k][:3])
        assert_close_scaled(res[k][0], Co, 1e-13, f"thread {k} C_tang")
        assert_close_scaled(res[k][2], dpo, 1e-13, f"thread {k} dp")


# ------------------------------------------------------------------------------------------------- dxo_mgpu_* (RCCL inside the library)
@pytest.mark.parametrize("form", ["single_process", "rank"])
def test_mgpu_world_of_one_matches_the_single_gpu_call(ctx, oracle, form):
    """One GPU is all a test box has: the communicator (ncclCommInitAll / ncclCommInitRank, world 1) comes up, the
    kernel writes its block into the full-length arrays, the gather modes are no-ops, results equal the oracle. The
    world > 1 arithmetic (block offsets, remote runs) is covered by the gloo tests of tests/test_sharding.py."""
    import torch

    from dolfinx_external_operator_amd import GATHER_COMPACT, GATHER_FULL, GATHER_NONE, MultiGpu

    n, d = 6400, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=81)
    if form == "single_process":
        g = MultiGpu(devices=[0])
    else:
        g = MultiGpu.from_rank(ctx, MultiGpu.unique_id(), 0, 1)
    try:
        assert g.world == 1 and g.local_count == 1 and g.rank(0) == 0
        g.set_stream(0, torch.cuda.current_stream().cuda_stream)
        t_in = [_dev(a) for a in (deps, sigma_n, p)]
        Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
        for gather in (GATHER_NONE, GATHER_FULL, GATHER_COMPACT):
            C = torch.full((n * d * d,), float("nan"), dtype=torch.float64, device="cuda:0")
            s = torch.full((n * d,), float("nan"), dtype=torch.float64, device="cuda:0")
            dp = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0")
            g.von_mises(PRM, d, n, gather, [t_in[0]], [t_in[1]], [t_in[2]], [C], [s], [dp])
            g.synchronize()
            assert_close_scaled(C.cpu().numpy(), Co, 1e-13, f"C_tang gather={gather}")
            assert_close_scaled(s.cpu().numpy(), so, 1e-13, "sigma")
            assert_close_scaled(dp.cpu().numpy(), dpo, 1e-13, "dp")
        # the collective itself, world 1: in place, data unchanged
        buf = torch.arange(1000, dtype=torch.float64, device="cuda:0")
        g.all_gather([buf], 1000)
        g.synchronize()
        assert torch.equal(buf, torch.arange(1000, dtype=torch.float64, device="cuda:0"))
        with pytest.raises(ValueError):
            g.von_mises(PRM, 5, n, GATHER_NONE, [t_in[0]], [t_in[1]], [t_in[2]], [C], [s], [dp])   # bad d
        with pytest.raises(ValueError):
            g.von_mises(PRM, d, n, GATHER_FULL, [t_in[0]], [t_in[1]], [t_in[2]], [C], [s], [])      # a pointer list without an entry for the device
    finally:
        g.close()


# ------------------------------------------------------------------------------------------------- float32 operands
def test_float32_operands_are_widened_and_results_narrowed(ctx, oracle, golden):
    """The reference's dispatcher is dtype-generic (test/test_multiaction.py:15-23 also runs float32); the kernels
    compute in fp64, float32 operands come back as float32 (demo_hyperelasticity.py:452-456: dtype follows the input)."""
    nc, nq, d = 64, 3, 4
    deps, sigma_n, p = vm_inputs(nc * nq, d, seed=12)
    ext = make_von_mises(sigma_n.astype(np.float32), p.astype(np.float32), ctx=ctx)
    C, s, dp = ext((1,))(deps.astype(np.float32).reshape(nc, nq, d))
    assert C.dtype == s.dtype == dp.dtype == np.float32
    Co, so, dpo = oracle.von_mises(deps.astype(np.float32).astype(np.float64), sigma_n.astype(np.float32).astype(np.float64),
                                   p.astype(np.float32).astype(np.float64))
    assert_close_scaled(C, Co.astype(np.float32), 1e-6, "C_tang f32")
    assert_close_scaled(s, so.astype(np.float32), 1e-6, "sigma f32")
    g = golden("icnn_isihara.npz")
    w = golden("icnn_isihara_weights.npz")
    icnn = make_icnn({k: w[k] for k in w.files}, ctx=ctx)
    dP, P = icnn((1,))(g["F"].astype(np.float32).reshape(-1, 2, 2))
    assert dP.dtype == np.float32
    assert np.max(np.abs(dP - g["dP_f32in"].reshape(-1))) <= 5e-6 * np.max(np.abs(g["dP_f32in"]))
    assert np.max(np.abs(P - g["P_f32in"].reshape(-1))) <= 5e-6 * np.max(np.abs(g["P_f32in"]))
    with pytest.raises(TypeError):
        ext((1,))(deps.astype(np.complex128).reshape(nc, nq, d))


@pytest.mark.parametrize("d", [4, 6])
def test_default_factory_path_reproduces_the_reference_nan_points(ctx, oracle, golden, d):
    """make_von_mises' default NumPy path rebuilds C_tang on the host from (sigma, dp). The reference has two NaN
    cases (demo_plasticity_von_mises.py:318-319): sigma_eq == 0 (sigma is NaN too: in the goldens) and f_elastic == 0
    EXACTLY, where only n_elas = s/sigma_eq * 0/0 and with it the tangent is NaN while sigma and dp = +0 are finite —
    no trace in (sigma, dp), so the kernel marks such points in the sign bit of dp and the host half writes the NaN
    tangent. An exact f_elastic == 0 needs H a power of two here: f = (sigma_eq - sigma_0) - H p with p = (sigma_eq -
    sigma_0) / H is then zero in every evaluation order."""
    g = golden(f"von_mises_d{d}.npz")
    deps = g["deps"].reshape(-1, 1, d)
    ext = make_von_mises(g["sigma_n"].reshape(-1), g["p"].reshape(-1), ctx=ctx)                 # host_tangent="rebuild"
    C, s, dp = ext((1,))(deps)
    assert np.isnan(g["C_tang"]).any(), "the golden holds the reference's sigma_eq == 0 point"
    assert_close_scaled(C, g["C_tang"], 1e-13, "C_tang (rebuilt on the host)")                    # includes the NaN pattern
    assert_close_scaled(s, g["sigma"], 1e-13, "sigma")
    assert_close_scaled(dp, g["dp"], 1e-13, "dp")
    # f_elastic == 0 exactly, between ordinary points
    n = 300
    e, sn, p = vm_inputs(n, d, seed=2)
    x = 300.0 / np.sqrt(1.5)
    for i in (0, 137, n - 1):
        e[i] = 0.0
        sn[i] = 0.0
        sn[i, 3] = x
        p[i] = (np.sqrt(1.5 * (x * x)) - SIGMA_0) / 512.0
    Co, so, dpo = oracle.von_mises(e, sn, p, H=512.0)
    assert np.isnan(Co[[0, 137, n - 1]]).all() and np.isfinite(so).all() and np.isfinite(dpo).all()
    outs = {}
    old_min = ctx.get_option("vm_rebuild_min_points")
    ctx.set_option("vm_rebuild_min_points", 0)
    for mode in ("rebuild", "copy"):
        f = make_von_mises(sn.reshape(-1), p, ctx=ctx, H=512.0, host_tangent=mode)((1,))
        outs[mode] = f(e.reshape(n, 1, d))
        Cg, sg, dpg = outs[mode]
        assert_close_scaled(Cg, Co, 1e-13, f"C_tang {mode}")                                      # NaN at exactly the three points
        assert_close_scaled(sg, so, 1e-13, f"sigma {mode}")
        assert np.array_equal(dpg == 0.0, dpo == 0.0) and not np.signbit(dpg[dpg == 0.0]).any()   # the mark does not leak out
    ctx.set_option("vm_rebuild_min_points", old_min)
    assert np.array_equal(outs["rebuild"][1], outs["copy"][1]) and np.array_equal(outs["rebuild"][2], outs["copy"][2])


def test_arena_calibration_retries_a_rejected_winner_bounded(ctx):
    """Round 6 (csrc/arena.hip): a calibration whose winner is below `placement_accept_pct` of the best rate the context ever kept
    for the same probe and size buys one more search, `placement_rounds` at most; the block kept is the head-to-head winner."""
    n, d = 3_300_000, 6
    keys = ("placement_candidates", "placement_rounds", "placement_accept_pct", "placement_standout_pct", "placement_cache")
    old = {k: ctx.get_option(k) for k in keys}
    try:
        ctx.set_option("placement_cache", 0)                 # every request below searches (no block retained from an earlier test is handed back)
        ctx.set_option("placement_candidates", 4)
        ctx.set_option("placement_standout_pct", 0)          # first calibration of the class: accepted whatever its crowd looks like
        a = ctx.vm_output_tensors(n, d)
        assert a[0].dxo_block.info["rounds"] == 1, a[0].dxo_block.info
        ctx.set_option("placement_accept_pct", 1000)         # nothing reaches ten times the record: every winner is rejected
        b = ctx.vm_output_tensors(n, d)
        info = b[0].dxo_block.info
        assert info["rounds"] == 3 and info["chosen_GBps"] > 1000.0, info
        ctx.set_option("placement_rounds", 1)
        c = ctx.vm_output_tensors(n, d)
        assert c[0].dxo_block.info["rounds"] == 1
        ctx.set_option("placement_accept_pct", 1)            # anything passes
        ctx.set_option("placement_rounds", 3)
        e = ctx.vm_output_tensors(n, d)
        assert e[0].dxo_block.info["rounds"] == 1
        ptrs = {t[0].data_ptr() for t in (a, b, c, e)}
        assert len(ptrs) == 4                                   # four live blocks, none handed out twice
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)


def test_a_freed_calibrated_block_is_handed_to_the_next_request_of_its_size(ctx):
    """Round 6 (csrc/arena.hip, cache_take): dxo_output_free keeps ONE calibrated block; the next request of exactly its size and probe gets that
    block back (info.rounds == 0, no search: milliseconds instead of seconds) after a re-timing; a request of another size, option
    placement_cache = 0, or a block that no longer reaches placement_accept_pct of the class record all lead to a fresh search."""
    import gc

    n, d = 3_300_000, 6
    keys = ("placement_candidates", "placement_cache", "placement_accept_pct", "placement_standout_pct", "placement_rounds")
    old = {k: ctx.get_option(k) for k in keys}

    def alloc(points=n):
        t = list(ctx.vm_output_tensors(points, d))
        return t, t[0].data_ptr(), dict(t[0].dxo_block.info)

    def free(holder):
        holder.clear()                 # the last references to the block's tensors
        gc.collect()

    try:
        ctx.set_option("placement_candidates", 4)
        ctx.set_option("placement_standout_pct", 0)
        ctx.set_option("placement_rounds", 1)
        ctx.set_option("placement_accept_pct", 50)           # a re-timed block within a factor two of the record is taken back
        ctx.set_option("placement_cache", 0)
        a, pa, ia = alloc()                                   # cache off: searches, and its free releases the block
        assert ia["rounds"] == 1
        free(a)
        ctx.set_option("placement_cache", 1)
        b, pb, ib = alloc()                                   # nothing retained yet: a search
        assert ib["rounds"] == 1 and ib["calibration_ms"] > 20.0
        free(b)                                               # retained
        c, pc, ic = alloc()
        assert pc == pb and ic["rounds"] == 0 and ic["chosen_GBps"] > 1000.0 and ic["calibration_ms"] < 0.5 * ib["calibration_ms"], (ib, ic)
        assert ic["tuned_blocks_per_cu"] > 0
        # the block works as any other: a call into it
        import torch
        g = torch.Generator(device="cuda").manual_seed(3)
        e = torch.empty(n * d, dtype=torch.float64, device="cuda").normal_(0.0, 3e-3, generator=g)
        sn = torch.empty(n * d, dtype=torch.float64, device="cuda").normal_(0.0, 100.0, generator=g)
        p = torch.zeros(n, dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.von_mises(PRM, d, n, MEM_DEVICE, e.data_ptr(), sn.data_ptr(), p.data_ptr(), c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr())
        torch.cuda.synchronize()
        assert bool(torch.isfinite(c[1]).all())
        free(c)                                               # retained again
        other, po, io = alloc(n + 64 * 1024)                  # another size: the retained block is released, a search runs
        assert io["rounds"] == 1
        free(other)                                           # ... and this one is retained now
        f, pf, i_f = alloc()                                  # the first size again: not the retained block's size -> a search
        assert i_f["rounds"] == 1
        free(f)
        ctx.set_option("placement_accept_pct", 1000)          # nothing re-times at ten times the record: the retained block is dropped
        h, ph, ih = alloc()
        assert ih["rounds"] == 1
        free(h)
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
        ctx.set_option("placement_cache", 0)                  # release whatever this test left retained ...
        try:
            free(list(ctx.vm_output_tensors(n, d)))
        finally:
            ctx.set_option("placement_cache", old["placement_cache"])


def test_vm_output_alloc_calibrates_with_the_kernel_itself(ctx, oracle):
    """dxo_vm_output_alloc: candidates are timed running vm_tile; the record names the probe, the kinds and the launch
    shape that was fastest on the block kept; a later dxo_von_mises into that block (which picks the shape up) gives the
    oracle's numbers, bit-identical to the same call into a plain allocation."""
    import torch

    n, d = 3_300_000, 6
    old = {k: ctx.get_option(k) for k in ("placement_candidates",)}
    ctx.set_option("placement_candidates", 4)
    try:
        C, s, dp = ctx.vm_output_tensors(n, d)
    finally:
        for k, v in old.items():
            ctx.set_option(k, v)
    info = C.dxo_block.info
    assert info["mode"] == "candidates" and info["probe"] == "vm_tile", info
    assert info["tuned_blocks_per_cu"] in (0, 32) and 0 <= info["chosen"] < info["candidates"] <= 5
    assert all(b > 1000.0 for b in info["probe_GBps"]) and info["chosen_GBps"] > 1000.0
    assert C.numel() == n * d * d and s.numel() == n * d and dp.numel() == n
    assert s.data_ptr() % 256 == 0 and dp.data_ptr() % 256 == 0 and s.data_ptr() >= C.data_ptr() + C.numel() * 8
    g = torch.Generator(device="cuda:0").manual_seed(6)
    deps = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 3e-3, generator=g)
    sigma_n = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 100.0, generator=g)
    p = torch.empty(n, dtype=torch.float64, device="cuda:0").normal_(0, 1e-3, generator=g).abs_()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    C2, s2, dp2 = (torch.empty_like(t) for t in (C, s, dp))
    ctx.von_mises(PRM, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C2.data_ptr(), s2.data_ptr(), dp2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(C, C2) and torch.equal(s, s2) and torch.equal(dp, dp2)      # the launch shape changes no value
    idx = torch.arange(0, n, 1009, device="cuda:0")
    Co, so, dpo = oracle.von_mises(deps[idx].cpu().numpy(), sigma_n[idx].cpu().numpy(), p[idx].cpu().numpy())
    assert_close_scaled(C.view(n, d * d)[idx].cpu().numpy(), Co, 1e-13, "C_tang in the kernel-calibrated block")
    assert_close_scaled(dp[idx].cpu().numpy(), dpo, 1e-13, "dp")
    small = ctx.vm_output_tensors(1000, 4)                                  # below placement_min_bytes: plain, no calibration
    assert small[0].dxo_block.info["chosen"] == -1 and small[0].numel() == 16000


def test_output_alloc_probed_runs_the_callers_kernel(ctx):
    """dxo_output_alloc_probed: the candidates are exercised by the caller's own launch (here dxo_heat through the Python
    binding, re-entering the context from the callback); the record says so; results in the block are the kernel's."""
    import torch

    n = 18_000_000                                                        # 64 B/point of outputs -> 1.15 GB (above placement_min_bytes)
    T = torch.rand(n, device="cuda:0", dtype=torch.float64) + 0.5
    sg = torch.randn(n, 2, device="cuda:0", dtype=torch.float64)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    calls = []

    def launch(ptrs, shape):
        calls.append(shape)
        ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), *ptrs)

    old = ctx.get_option("placement_candidates")
    ctx.set_option("placement_candidates", 4)
    try:
        q, dT, ds = ctx.output_tensors_probed((n * 2, n * 2, n * 4), launch, bytes_per_launch=88.0 * n)
    finally:
        ctx.set_option("placement_candidates", old)
    info = q.dxo_block.info
    assert info["probe"] == "caller" and info["mode"] == "candidates" and len(calls) >= 4 * 5 and set(calls) == {0}, (info, len(calls))
    assert all(b > 500.0 for b in info["probe_GBps"])
    ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), q.data_ptr(), dT.data_ptr(), ds.data_ptr())
    torch.cuda.synchronize()
    k = 1.0 / (1.0 + T)
    assert torch.allclose(q.view(n, 2), -k[:, None] * sg, rtol=1e-14, atol=0) and torch.allclose(dT.view(n, 2), (k * k)[:, None] * sg, rtol=1e-14, atol=0)

    def bad(ptrs, shape):
        raise RuntimeError("boom")

    with pytest.raises(RuntimeError, match="boom"):                       # an exception in the callback surfaces after the call
        ctx.output_tensors_probed((n * 8,), bad)


def test_outputs_land_in_the_callers_coefficient_storage(ctx, oracle):
    """`outputs=`: the kernel's results go straight into arrays the caller owns (the operator's coefficient, the Functions
    the demo copies the extras into) and those very arrays come back — the reference's `x.array[:] = values`
    (external_operator.py:289-290) then assigns an array to itself, which NumPy skips. Checked through the dispatcher
    mirror, with pageable and with page-locked (Context.pin) targets, rebuild and copy mode."""
    from dolfinx_external_operator_amd import QuadratureExternalOperator, evaluate_external_operators, evaluate_operands
    from dolfinx_external_operator_amd.evaluation import Operand

    nc, nq, d = 40_000, 8, 6
    n = nc * nq                                                       # 320 000 points: above the rebuild threshold
    deps, sigma_n, p = vm_inputs(n, d, seed=91)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    for mode in ("rebuild", "copy"):
        for pinned in (False, True):
            sig_store, dp_store = np.full(n * d, np.nan), np.full(n, np.nan)
            op = QuadratureExternalOperator(Operand(lambda cells: deps.reshape(nc, nq, d)[cells], "deps"), num_cells=nc, num_points=nq,
                                            value_shape=(d, d), derivatives=(1,))
            coeff = op.ref_coefficient.x.array
            if pinned:
                ctx.pin(coeff)
            try:
                op.external_function = make_von_mises(sigma_n, p, ctx=ctx, host_tangent=mode, outputs=(op.ref_coefficient, sig_store, dp_store))
                ((C, s, dp),) = evaluate_external_operators([op], evaluate_operands([op]))
                assert np.shares_memory(C, coeff) and C.ctypes.data == coeff.ctypes.data      # the assignment was array-to-itself
                assert s.ctypes.data == sig_store.ctypes.data and dp.ctypes.data == dp_store.ctypes.data
                assert_close_scaled(coeff, Co.reshape(-1), 1e-13, f"coefficient storage ({mode}, pinned={pinned})")
                assert_close_scaled(sig_store, so.reshape(-1), 1e-13, "sigma store")
                assert_close_scaled(dp_store, dpo.reshape(-1), 1e-13, "dp store")
            finally:
                if pinned:
                    ctx.unpin(coeff)
    with pytest.raises(ValueError, match="entries"):                   # a target of the wrong size is refused, nothing is written
        make_von_mises(sigma_n, p, ctx=ctx, outputs=(np.zeros(10), None, None))((1,))(deps.reshape(nc, nq, d))
    with pytest.raises(TypeError, match="float64"):
        make_von_mises(sigma_n, p, ctx=ctx, outputs=(np.zeros(n * d * d, dtype=np.float32), None, None))((1,))(deps.reshape(nc, nq, d))


def test_outputs_targets_of_the_other_factories(ctx, oracle, golden):
    """`outputs=` on make_mohr_coulomb, make_icnn, make_isihara: results land in the caller's arrays, same bits as the
    default call."""
    n = 3000
    deps, sigma_n = mc_tracing_inputs(oracle, n, seed=23)
    C0, s0 = make_mohr_coulomb(sigma_n, ctx=ctx, diagnostics=False)((1,))(deps)
    C_t, s_t = np.full(n * 16, np.nan), np.full(n * 4, np.nan)
    C1, s1 = make_mohr_coulomb(sigma_n, ctx=ctx, diagnostics=False, outputs=(C_t, s_t))((1,))(deps)
    assert C1.ctypes.data == C_t.ctypes.data and s1.ctypes.data == s_t.ctypes.data
    assert np.array_equal(C_t, C0) and np.array_equal(s_t, s0)
    g, w = golden("icnn_isihara.npz"), golden("icnn_isihara_weights.npz")
    F = np.asarray(g["F"], dtype=np.float64).reshape(-1, 2, 2)
    m = F.shape[0]
    for make in (lambda **kw: make_icnn({k: w[k] for k in w.files}, ctx=ctx, **kw), lambda **kw: make_isihara(ctx=ctx, **kw)):
        dP0, P0 = make()((1,))(F)
        dP_t, P_t = np.full(m * 16, np.nan), np.full(m * 4, np.nan)
        dP1, P1 = make(outputs=(dP_t, None))((1,))(F)                  # only the tangent has a target; P is a fresh array
        assert dP1.ctypes.data == dP_t.ctypes.data and P1.ctypes.data != P_t.ctypes.data
        assert np.array_equal(dP_t, dP0) and np.array_equal(P1, P0) and np.isnan(P_t).all()


def test_vm_output_alloc_d4_reference_layout(ctx, oracle):
    """The reference demo's layout (d = 4) in a kernel-calibrated block: 8*10^6 points = 1.34 GB of outputs, tuned launch
    shape picked up by the call, oracle on a strided sample, every entry written."""
    import torch

    n, d = 8_000_000, 4
    old = ctx.get_option("placement_candidates")
    ctx.set_option("placement_candidates", 4)
    try:
        C, s, dp = ctx.vm_output_tensors(n, d)
    finally:
        ctx.set_option("placement_candidates", old)
    assert C.dxo_block.info["probe"] == "vm_tile" and C.numel() == n * 16
    g = torch.Generator(device="cuda:0").manual_seed(9)
    deps = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 3e-3, generator=g)
    sigma_n = torch.empty(n, d, dtype=torch.float64, device="cuda:0").normal_(0, 100.0, generator=g)
    p = torch.empty(n, dtype=torch.float64, device="cuda:0").normal_(0, 1e-3, generator=g).abs_()
    C.fill_(float("nan")); s.fill_(float("nan")); dp.fill_(float("nan"))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.von_mises(PRM, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(C).all()) and bool(torch.isfinite(s).all()) and bool(torch.isfinite(dp).all())
    idx = torch.cat([torch.arange(0, n, 2003, device="cuda:0"), torch.tensor([n - 1], device="cuda:0")])
    Co, so, dpo = oracle.von_mises(deps[idx].cpu().numpy(), sigma_n[idx].cpu().numpy(), p[idx].cpu().numpy())
    assert_close_scaled(C.view(n, 16)[idx].cpu().numpy(), Co, 1e-13, "C_tang d=4")
    assert_close_scaled(s.view(n, d)[idx].cpu().numpy(), so, 1e-13, "sigma d=4")
    assert_close_scaled(dp[idx].cpu().numpy(), dpo, 1e-13, "dp d=4")
