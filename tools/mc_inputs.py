"""Synthetic Mohr-Coulomb inputs of BASELINE config 4: the demo's yield-surface tracing distribution
(doc/demo/demo_plasticity_mohr_coulomb.py:854-929). Used by the tests (through tests/conftest.py), scripts/bench_mc.py and
bench.py's secondary block. `oracle` is the CPU checker (oracle/mc_oracle.cpp): it only advances the STATES along the
tracing loads so that the batch mixes elastic points and points sitting on the yield surface; nothing measured calls it."""
from __future__ import annotations

import numpy as np

MC_E, MC_NU = 6778.0, 0.25  # demo_plasticity_mohr_coulomb.py:110-111


def mc_elastic_matrices():
    lm = MC_E * MC_NU / ((1 + MC_NU) * (1 - 2 * MC_NU))
    mu = MC_E / (2 * (1 + MC_NU))
    C = np.array([[lm + 2 * mu, lm, lm, 0], [lm, lm + 2 * mu, lm, 0], [lm, lm, lm + 2 * mu, 0], [0, 0, 0, 2 * mu]])
    return C, np.linalg.inv(C)


def mc_path_increment(theta, R):
    """Stress increment of the demo's yield-surface tracing (demo_plasticity_mohr_coulomb.py:868-871)."""
    d = np.zeros((len(theta), 4))
    d[:, 0] = (R / np.sqrt(2)) * (np.cos(theta) + np.sin(theta) / np.sqrt(3))
    d[:, 1] = (R / np.sqrt(2)) * (-2 * np.sin(theta) / np.sqrt(3))
    d[:, 2] = (R / np.sqrt(2)) * (np.sin(theta) / np.sqrt(3) - np.cos(theta))
    return d


def mc_tracing_inputs(oracle, n, seed, shear=0.0):
    """SURVEY.md 8(d) config 4 distribution: random Lode angle theta ~ U(-pi/6, pi/6), states after
    k in {0..8} tracing loads of R = 0.7 from the hydrostatic state p = 0.1 (:854-929), then an increment of
    R ~ U(0, 0.7) along the same path. `shear` > 0 adds a Mandel shear component to state and increment.
    Returns deps (n,4), sigma_n (n,4)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    _, S = mc_elastic_matrices()
    tr = np.array([1.0, 1.0, 1.0, 0.0])
    theta = rng.uniform(-np.pi / 6 + 1e-5, np.pi / 6 - 1e-5, n)
    k_loads = rng.integers(0, 9, n)
    sn = np.zeros((n, 4))
    sn[:, :3] = 0.1
    if shear > 0:
        sn[:, 3] = rng.normal(0, shear, n)
    for k in range(8):
        active = k_loads > k
        if not active.any():
            break
        d = mc_path_increment(theta[active], 0.7)
        _, s, *_ = oracle.mohr_coulomb(d @ S.T, sn[active], nthreads=8, tangent=False)
        dp = s @ tr / 3.0 - 0.1
        sn[active] = s - np.outer(dp, tr)          # :922-923
    dsig = mc_path_increment(theta, rng.uniform(0.0, 0.7, n))
    if shear > 0:
        dsig[:, 3] = rng.normal(0, shear, n)
    return dsig @ S.T, sn


def mc_default_params():
    """dxo_mc_params of the demo (demo_plasticity_mohr_coulomb.py:110-116, 469)."""
    from dolfinx_external_operator_amd import McParams

    c, phi = 3.45, np.pi / 6
    return McParams(MC_E, MC_NU, c, phi, phi, 26 * np.pi / 180, 0.26 * c / np.tan(phi), 1e-8, 200, 0)


def mc_tracing_inputs_device(ctx, n, seed, pool=50_000):
    """The same distribution produced WITHOUT the CPU checker: the states after k tracing loads are advanced by the
    library's own kernel (dxo_mohr_coulomb on device memory, `pool` seeded points x 8 loads), then `n` points are drawn
    from the pool with the increment scaled by U(0.5, 1). Returns torch CUDA tensors deps (n,4), sigma_n (n,4).
    This is what bench.py's secondary block times (no oracle call outside its cpu_baseline leg)."""
    import torch

    from dolfinx_external_operator_amd import MEM_DEVICE

    dev = torch.device("cuda", ctx.device)
    prm = mc_default_params()
    rng = np.random.Generator(np.random.PCG64(seed))
    _, S = mc_elastic_matrices()
    St = torch.from_numpy(S.T.copy()).to(dev)
    tr = torch.tensor([1.0, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
    theta = rng.uniform(-np.pi / 6 + 1e-5, np.pi / 6 - 1e-5, pool)
    k_loads = torch.from_numpy(rng.integers(0, 9, pool)).to(dev)
    sn = torch.zeros(pool, 4, dtype=torch.float64, device=dev)
    sn[:, :3] = 0.1
    d_full = torch.from_numpy(mc_path_increment(theta, 0.7)).to(dev)
    for k in range(8):
        idx = torch.nonzero(k_loads > k).squeeze(1)
        m = int(idx.numel())
        if m == 0:
            break
        de = (d_full[idx] @ St).contiguous()
        s_in = sn[idx].contiguous()
        Ct = torch.empty(m * 16, dtype=torch.float64, device=dev)
        s_out = torch.empty(m, 4, dtype=torch.float64, device=dev)
        ctx.mohr_coulomb(prm, m, MEM_DEVICE, de.data_ptr(), s_in.data_ptr(), Ct.data_ptr(), s_out.data_ptr())
        ctx.synchronize()
        dpv = s_out @ tr / 3.0 - 0.1
        sn[idx] = s_out - torch.outer(dpv, tr)          # :922-923
    dsig = torch.from_numpy(mc_path_increment(theta, rng.uniform(0.0, 0.7, pool))).to(dev)
    pool_d = dsig @ St
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pick = torch.randint(0, pool, (n,), generator=g, device=dev)
    scale = torch.rand(n, 1, generator=g, device=dev, dtype=torch.float64) * 0.5 + 0.5
    return (pool_d[pick] * scale).contiguous(), sn[pick].contiguous()
