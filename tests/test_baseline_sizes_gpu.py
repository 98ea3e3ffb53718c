"""The kernels at BASELINE.json's full sizes inside `pytest -m gpu`.

Each test runs ONE call at the configuration's size on device-resident inputs, compares a strided sample of 10^5
points with the CPU oracle (the oracle finishes that in seconds), and checks size-independent properties over the
WHOLE output on the device:
  config 2 / north star  von Mises d = 6, 10^7 points: yield condition on plastic points, C_elas bit-exact on elastic
                         points, tangent symmetry, finiteness
  config 4               Mohr-Coulomb, 10^7 points of the yield-surface tracing distribution: elastic
                         points -> one iteration and C_elas exactly, converged plastic points sit on f = 0 with
                         dlambda > 0, iteration histogram equal to the oracle's on the sample
  config 5               ICNN, 10^6 points F = I + 0.1 N(0,1) with det F > 0.2: fp32-network parity with the oracle,
                         finiteness, tangent major symmetry at fp32 noise level
Tolerances are the ones of the small-size parity tests (1e-13 x scale von Mises; mc_compare's for Mohr-Coulomb;
2e-6 x scale ICNN).
"""
import numpy as np
import pytest

from conftest import assert_close_scaled, mc_compare_all, mc_elastic_matrices
from dolfinx_external_operator_amd import MEM_DEVICE, McParams, VmParams

pytestmark = pytest.mark.gpu

E, NU, SIGMA_0 = 70e3, 0.3, 250.0
H = E * (E / 100.0) / (E - E / 100.0)


def _strided(n, m, device):
    import torch

    idx = torch.arange(0, n, max(n // m, 1), device=device)[:m]
    return torch.cat([idx, torch.tensor([n - 1], device=device)])


def test_von_mises_d6_at_ten_million_points(ctx, oracle):
    import torch

    n, d = 10_000_000, 6
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    deps = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 3e-3, generator=g)
    deps[:, 3:] *= 2.0 ** 0.5
    sigma_n = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 100.0, generator=g)
    p = torch.empty(n, dtype=torch.float64, device=dev).normal_(0.0, 1e-3, generator=g).abs_()
    C, s, dp = ctx.vm_output_tensors(n, d)                          # the kernel-calibrated arena block, as the bench uses it
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = VmParams(E, NU, SIGMA_0, H)
    ctx.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())
    torch.cuda.synchronize()
    # -- oracle on a strided sample of 10^5 points (+ the last point)
    idx = _strided(n, 100_000, dev)
    Co, so, dpo = oracle.von_mises(deps[idx].cpu().numpy(), sigma_n[idx].cpu().numpy(), p[idx].cpu().numpy(), nthreads=8)
    assert_close_scaled(C.view(n, d * d)[idx].cpu().numpy(), Co, 1e-13, "C_tang sample")
    assert_close_scaled(s.view(n, d)[idx].cpu().numpy(), so, 1e-13, "sigma sample")
    assert_close_scaled(dp[idx].cpu().numpy(), dpo, 1e-13, "dp sample")
    # -- whole-array properties, on the device
    S = s.view(n, d)
    dev_s = S.clone()
    dev_s[:, :3] -= S[:, :3].mean(dim=1, keepdim=True)
    f = (1.5 * (dev_s * dev_s).sum(1)).sqrt() - SIGMA_0 - H * (p + dp)
    plastic = dp > 0
    frac = float(plastic.double().mean())
    assert 0.5 < frac < 0.999, frac                                    # mostly plastic, with elastic points present (measured 97 %)
    assert float(f[plastic].abs().max()) <= 1e-8 * SIGMA_0             # plastic points sit on the yield surface
    assert float(f[~plastic].max()) <= 1e-9 * SIGMA_0                  # elastic points are inside it
    lm, mu = E * NU / (1 + NU) / (1 - 2 * NU), E / 2 / (1 + NU)
    C_el = torch.zeros(d, d, dtype=torch.float64, device=dev)
    C_el[:3, :3] = lm
    C_el += 2 * mu * torch.eye(d, dtype=torch.float64, device=dev)
    Cv = C.view(n, d, d)
    assert bool((Cv[~plastic] == C_el).all())                          # elastic points: C_elas bit for bit
    assert bool(torch.isfinite(C).all()) and bool(torch.isfinite(s).all())
    asym = 0.0
    for lo in range(0, n, 2_000_000):                                  # in slices: a full transpose would double 2.9 GB
        blk = Cv[lo:lo + 2_000_000]
        asym = max(asym, float((blk - blk.transpose(1, 2)).abs().max()))
    assert asym <= 1e-9 * E


def test_mohr_coulomb_at_ten_million_points(ctx, oracle):
    import torch

    n = 10_000_000
    from tools.mc_inputs import mc_pool_inputs

    deps, sn = mc_pool_inputs(n, seed=0)      # drawn from the frozen pool (tests/golden/mc_tracing_pool.npz): inputs need no checker
    dev = torch.device("cuda:0")
    d_deps, d_sn = torch.from_numpy(deps).to(dev), torch.from_numpy(sn).to(dev)
    Ct = torch.empty(n * 16, dtype=torch.float64, device=dev)
    s = torch.empty(n * 4, dtype=torch.float64, device=dev)
    it = torch.empty(n, dtype=torch.int32, device=dev)
    y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    c = 3.45
    phi = 30 * np.pi / 180
    prm = McParams(6778.0, 0.25, c, phi, phi, 26 * np.pi / 180, 0.26 * c / np.tan(phi), 1e-8, 200, 0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.mohr_coulomb(prm, n, MEM_DEVICE, d_deps.data_ptr(), d_sn.data_ptr(), Ct.data_ptr(), s.data_ptr(), it.data_ptr(),
                     y.data_ptr(), nr.data_ptr(), dl.data_ptr())
    torch.cuda.synchronize()
    # -- oracle on a strided sample
    sel = _strided(n, 100_000, dev)
    sel_h = sel.cpu().numpy()
    ref = oracle.mohr_coulomb(deps[sel_h], sn[sel_h], nthreads=8)
    got = (Ct.view(n, 4, 4)[sel].cpu().numpy(), s.view(n, 4)[sel].cpu().numpy(), it[sel].cpu().numpy(), y[sel].cpu().numpy(),
           nr[sel].cpu().numpy(), dl[sel].cpu().numpy())
    mc_compare_all(got, ref, "10^7-point call, strided sample vs oracle", sn[sel_h])
    # -- the summary the reference prints at every call (:584-591), over all 10^7 points, reduced on the GPU
    summ = ctx.mc_summary(n, it, y, nr, nbins=201)
    assert int(summ["counts"].sum()) == n
    u_ref, c_ref = np.unique(ref[2], return_counts=True)                # same iteration classes as the oracle's sample
    assert set(u_ref.tolist()) <= set(summ["unique_iters"].tolist())
    dead = it >= 200
    assert float(dead.double().mean()) < 0.01                           # tracing distribution: (nearly) every point converges
    assert float(nr[~dead & (y > 0)].max()) <= 1e-8                     # ... to the reference's tolerance (:469)
    # -- whole-array properties
    el = y <= 0
    frac_el = float(el.double().mean())
    assert 0.2 < frac_el < 0.9, frac_el
    Cel = torch.from_numpy(mc_elastic_matrices()[0]).to(dev)
    assert bool((it[el] == 1).all()) and bool((Ct.view(n, 4, 4)[el] == Cel).all())
    assert bool((dl[~el & ~dead] > 0).all()) and bool((dl[el] == 0).all())
    assert bool(torch.isfinite(Ct).all()) and bool(torch.isfinite(s).all())
    pl = torch.nonzero(~el & ~dead).reshape(-1)
    pick = pl[:: max(pl.numel() // 50_000, 1)][:50_000]
    f_after = oracle.mc_surface(s.view(n, 4)[pick].cpu().numpy())[0]
    assert np.max(np.abs(f_after)) < 1e-6                              # returned stresses lie on the yield surface


def test_icnn_at_one_million_points(ctx, golden):
    import torch

    from dolfinx_external_operator_amd import make_icnn
    from oracle.icnn_oracle import icnn_stress_tangent

    w = dict(golden("icnn_isihara_weights.npz"))
    n = 1_000_000
    rng = np.random.Generator(np.random.PCG64(3))
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(int(n * 1.2), 4))
    F = F[(F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]) > 0.2][:n]          # det F > 0.2
    assert F.shape[0] == n
    ext = make_icnn({k.replace("__", "."): v for k, v in w.items()}, ctx=ctx)
    dP, P = ext((1,))(torch.from_numpy(F).to("cuda:0").reshape(n, 1, 2, 2))
    torch.cuda.synchronize()
    sel = np.arange(0, n, 10)                                           # 10^5 points
    dPo, Po = icnn_stress_tangent(F[sel], w)
    sel_d = torch.from_numpy(sel).to("cuda:0")
    got_dP, got_P = dP.view(n, 16)[sel_d].cpu().numpy(), P.view(n, 4)[sel_d].cpu().numpy()
    assert np.max(np.abs(got_dP - dPo.reshape(-1, 16))) <= 2e-6 * np.max(np.abs(dPo))
    assert np.max(np.abs(got_P - Po)) <= 2e-6 * np.max(np.abs(Po))
    assert bool(torch.isfinite(dP).all()) and bool(torch.isfinite(P).all())
    T = dP.view(n, 4, 4)
    assert float((T - T.transpose(1, 2)).abs().max()) <= 2e-5 * float(T.abs().max())   # hyperelastic tangent: major symmetry, fp32 noise


def test_icnn_kernels_agree_at_ten_million_points(ctx, golden):
    """Config 5's upper size: the default kernel (GEMMs as split-bf16 products) against the fp32-input MFMA kernel over ALL 10^7
    points on the device, and a strided sample of both against the oracle."""
    import torch

    from oracle.icnn_oracle import icnn_stress_tangent

    w = dict(golden("icnn_isihara_weights.npz"))
    n = 10_000_000
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(4)
    eye = torch.tensor([1.0, 0.0, 0.0, 1.0], device=dev, dtype=torch.float64)
    F = torch.randn(n, 4, device=dev, dtype=torch.float64, generator=g) * 0.1 + eye
    F[(F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]) <= 0.2] = eye
    model = ctx.icnn_create({k.replace("__", "."): v for k, v in w.items()})
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    out = {}
    try:
        for variant in (2, 1):
            ctx.set_option("icnn_variant", variant)
            dP = torch.empty(n * 16, device=dev, dtype=torch.float64)
            P = torch.empty(n * 4, device=dev, dtype=torch.float64)
            ctx.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr())
            out[variant] = (dP, P)
        torch.cuda.synchronize()
    finally:
        ctx.set_option("icnn_variant", 2)
        ctx.icnn_destroy(model)
    sd, sp = float(out[1][0].abs().max()), float(out[1][1].abs().max())
    assert float((out[2][0] - out[1][0]).abs().max()) <= 2e-6 * sd and float((out[2][1] - out[1][1]).abs().max()) <= 2e-6 * sp
    assert bool(torch.isfinite(out[2][0]).all()) and bool(torch.isfinite(out[2][1]).all())
    sel = _strided(n, 50_000, dev)
    dPo, Po = icnn_stress_tangent(F[sel].cpu().numpy(), w)
    for variant in (2, 1):
        got_dP = out[variant][0].view(n, 16)[sel].cpu().numpy()
        got_P = out[variant][1].view(n, 4)[sel].cpu().numpy()
        assert np.max(np.abs(got_dP - dPo.reshape(-1, 16))) <= 2e-6 * np.max(np.abs(dPo))
        assert np.max(np.abs(got_P - Po)) <= 2e-6 * np.max(np.abs(Po))


def test_mohr_coulomb_schedules_agree_bitwise_at_ten_million_points(ctx, oracle):
    """Config 4's size: the single persistent kernel (default) and the classify + Newton pair run the same per-point arithmetic
    in different orders on different lanes — every one of the 10^7 x 24 outputs is the same bit pattern."""
    import torch

    from tools.mc_inputs import mc_default_params, mc_pool_inputs_device

    n = 10_000_000
    dev = torch.device("cuda:0")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    prm = mc_default_params()
    deps, sn = mc_pool_inputs_device(torch, dev, n, seed=5)
    out = {}
    try:
        for variant in (2, 1):
            ctx.set_option("mc_variant", variant)
            Ct = torch.empty(n * 16, dtype=torch.float64, device=dev)
            s = torch.empty(n * 4, dtype=torch.float64, device=dev)
            it = torch.empty(n, dtype=torch.int32, device=dev)
            y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
            ctx.mohr_coulomb(prm, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), Ct.data_ptr(), s.data_ptr(), it.data_ptr(),
                             y.data_ptr(), nr.data_ptr(), dl.data_ptr())
            out[variant] = (Ct, s, it, y, nr, dl)
        torch.cuda.synchronize()
    finally:
        ctx.set_option("mc_variant", 2)
    assert 0.1 < float((out[2][3] > 0).double().mean()) < 0.9           # a real mix of elastic and plastic points
    for a, b in zip(out[2], out[1]):
        if a.dtype == torch.int32:
            assert bool(torch.equal(a, b))
        else:                                                           # bit patterns, so that equal NaNs count as equal
            assert bool(torch.equal(a.view(torch.int64), b.view(torch.int64)))


def test_config_3_cell_blocks_at_a_hundred_million_points_on_one_gpu(ctx, oracle):
    """BASELINE config 3 (von Mises, 12.5 * 10^6 hexahedra x 8 points = 10^8 points, cell-block sharded over 8 GPUs) with
    the eight blocks run ONE AFTER ANOTHER on the single GPU of this box, each writing its slice of the full-length
    arrays exactly as a rank does before the gather (sharding.CellBlockPartition gives the ranges; the compact gather's
    tangent rebuild, dxo_vm_expand_tangent over the 'remote' ranges of rank 0, is run as well). What this pins at the
    configuration's real size: the block offsets (34 GB of outputs, indices beyond 2^32), bit-identity of a block's slice
    with one whole-array call, the oracle on a strided sample, and the rebuilt remote tangents against the owners'.
    The exchange itself needs the 8-GPU node."""
    import torch

    from dolfinx_external_operator_amd.sharding import CellBlockPartition, remote_point_ranges

    world, nq, d = 8, 8, 6
    part = CellBlockPartition(12_500_000, nq, world)
    n_rank, N = part.points_per_rank, part.padded_points
    assert part.num_points == 100_000_000 and n_rank % 64 == 0 and 0 <= N - part.num_points < world * 64   # blocks end on wave tiles
    dev = torch.device("cuda:0")
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b < 110 * 2**30:
        pytest.skip("needs ~100 GB of free HBM")
    g = torch.Generator(device=dev).manual_seed(100)
    deps = torch.empty(N, d, dtype=torch.float64, device=dev).normal_(0.0, 3e-3, generator=g)
    sigma_n = torch.empty(N, d, dtype=torch.float64, device=dev).normal_(0.0, 100.0, generator=g)
    p = torch.empty(N, dtype=torch.float64, device=dev).normal_(0.0, 1e-3, generator=g).abs_()
    C = torch.full((N * d * d,), float("nan"), dtype=torch.float64, device=dev)
    s = torch.full((N * d,), float("nan"), dtype=torch.float64, device=dev)
    dp = torch.full((N,), float("nan"), dtype=torch.float64, device=dev)
    prm = VmParams(E, NU, SIGMA_0, H)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for rank in range(world):                                          # what rank `rank` launches
        rb, re = part.point_range(rank)                              # the rank's REAL points; its block is padded to n_rank
        b, e = rank * n_rank, (rank + 1) * n_rank
        assert rb == b and b < re <= e and (re == e or rank == world - 1)
        ctx.von_mises(prm, d, e - b, MEM_DEVICE, deps[b:e].data_ptr(), sigma_n[b:e].data_ptr(), p[b:e].data_ptr(),
                      C[b * d * d:].data_ptr(), s[b * d:].data_ptr(), dp[b:].data_ptr())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(s).all()) and bool(torch.isfinite(dp).all())           # every slice was written
    assert bool(torch.isfinite(C[-4096 * d * d:]).all()) and bool(torch.isfinite(C[: 4096 * d * d]).all())
    # one whole-array call gives the same bits as the eight block calls
    s2, dp2 = torch.empty_like(s), torch.empty_like(dp)
    C2 = torch.empty(n_rank * d * d, dtype=torch.float64, device=dev)                  # tangent of the LAST block only (memory)
    last = (world - 1) * n_rank
    ctx.von_mises(prm, d, n_rank, MEM_DEVICE, deps[last:].data_ptr(), sigma_n[last:].data_ptr(), p[last:].data_ptr(),
                  C2.data_ptr(), s2[last * d:].data_ptr(), dp2[last:].data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(C2, C[last * d * d:]) and torch.equal(s2[last * d:], s[last * d:]) and torch.equal(dp2[last:], dp[last:])
    del C2, s2, dp2
    # oracle on a strided sample across all blocks (+ the first and last point of every block)
    idx = torch.cat([torch.arange(0, N, 1009, device=dev), torch.arange(0, N, n_rank, device=dev), torch.arange(n_rank - 1, N, n_rank, device=dev)])
    Co, so, dpo = oracle.von_mises(deps[idx].cpu().numpy(), sigma_n[idx].cpu().numpy(), p[idx].cpu().numpy(), nthreads=8)
    assert_close_scaled(C.view(N, d * d)[idx].cpu().numpy(), Co, 1e-13, "C_tang sample over the eight blocks")
    assert_close_scaled(s.view(N, d)[idx].cpu().numpy(), so, 1e-13, "sigma sample")
    assert_close_scaled(dp[idx].cpu().numpy(), dpo, 1e-13, "dp sample")
    # rank 0 after a compact gather: the tangents of the remote ranges rebuilt from (sigma, dp) agree with the owners'
    for (b, e) in remote_point_ranges(0, world, n_rank):
        owner = C[b * d * d: b * d * d + 4096 * d * d].clone()
        tail = C[e * d * d - 4096 * d * d: e * d * d].clone()
        ctx.vm_expand_tangent(prm, d, e - b, MEM_DEVICE, s[b * d:].data_ptr(), dp[b:].data_ptr(), C[b * d * d:].data_ptr())
        torch.cuda.synchronize()
        scale = float(owner.abs().max())
        assert float((C[b * d * d: b * d * d + 4096 * d * d] - owner).abs().max()) <= 1e-13 * scale
        assert float((C[e * d * d - 4096 * d * d: e * d * d] - tail).abs().max()) <= 1e-13 * scale
