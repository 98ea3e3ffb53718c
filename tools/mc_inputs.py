"""Synthetic Mohr-Coulomb inputs of BASELINE config 4: the demo's yield-surface tracing distribution
(doc/demo/demo_plasticity_mohr_coulomb.py:854-929). Used by scripts/bench_mc.py, bench.py's secondary block and the 10^7-point
test. The states of that distribution (stress after k tracing loads) need a return map to produce: that was done ONCE, by the CPU
checker, into the fixture tests/golden/mc_tracing_pool.npz (tests/golden/make_golden_mc_inputs.py); this module only reads the
fixture — it never imports or calls anything under oracle/. (The tests' own seeded generator, which does call the checker, lives
in tests/conftest.py: mc_tracing_inputs.)"""
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


_POOL = None


def mc_pool():
    """(deps, sigma_n), each (20 000, 4): the frozen pool of the tracing distribution (tests/golden/mc_tracing_pool.npz, written once by
    tests/golden/make_golden_mc_inputs.py: the states after k tracing loads need a return map, which the CPU checker did THERE). Nothing
    in this module calls the checker; bench legs, scripts and the 10^7-point test draw from this pool."""
    global _POOL
    if _POOL is None:
        import pathlib

        g = np.load(pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden" / "mc_tracing_pool.npz")
        _, S = mc_elastic_matrices()
        sn = np.zeros((g["theta"].size, 4))
        sn[:, :3] = g["sigma_n3"]
        _POOL = (mc_path_increment(g["theta"], g["R"]) @ S.T, sn)      # deps = S_elas dsigma (:903)
    return _POOL


def mc_pool_inputs(n, seed=0):
    """n points of BASELINE config 4 drawn from the frozen pool (NumPy, host): a seeded pick with the increment scaled by U(0.5, 1)."""
    pool_d, pool_s = mc_pool()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, pool_d.shape[0], n)
    return pool_d[idx] * rng.uniform(0.5, 1.0, (n, 1)), pool_s[idx]


def mc_pool_inputs_device(torch, device, n, seed=2):
    """The same draw made on the device (CUDA tensors deps (n, 4), sigma_n (n, 4)): what the bench leg of config 4 times."""
    pool_d, pool_s = mc_pool()
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pick = torch.randint(0, pool_d.shape[0], (n,), generator=g, device=device)
    scale = torch.rand(n, 1, generator=g, device=device, dtype=torch.float64) * 0.5 + 0.5
    return (torch.from_numpy(pool_d).to(device)[pick] * scale).contiguous(), torch.from_numpy(pool_s).to(device)[pick].contiguous()


def mc_default_params():
    """dxo_mc_params of the demo (demo_plasticity_mohr_coulomb.py:110-116, 469)."""
    from dolfinx_external_operator_amd import McParams

    c, phi = 3.45, np.pi / 6
    return McParams(MC_E, MC_NU, c, phi, phi, 26 * np.pi / 180, 0.26 * c / np.tan(phi), 1e-8, 200, 0)
