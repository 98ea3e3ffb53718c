#!/usr/bin/env python3
"""Freeze the inputs of BASELINE config 4 (Mohr-Coulomb, yield-surface tracing distribution) as a fixture.

    python tests/golden/make_golden_mc_inputs.py          ->  tests/golden/mc_tracing_pool.npz

The pool holds 20 000 seeded points of the demo's tracing experiment (doc/demo/demo_plasticity_mohr_coulomb.py:854-929): a Lode
angle theta ~ U(-pi/6, pi/6), the stress state after k in {0..8} tracing loads of R = 0.7 from the hydrostatic state p = 0.1
(projected back to p = 0.1 after every load, :922-923) and an increment of R ~ U(0, 0.7) along the same path (:868-871), as a
strain increment deps = S_elas dsigma (:903). Advancing the states needs a return map; the CPU checker (oracle/mc_oracle.cpp)
does that HERE, once — bench legs, scripts and the 10^7-point test then draw their inputs from this file (tools/mc_inputs.py:
mc_pool / mc_pool_inputs) and nothing under tools/ calls the checker any more. Arrays: theta (n,), k_loads (n,), R (n,),
sigma_n3 (n, 3) — the normal components of the states, the Mandel shear component is zero — ; deps is a closed-form function of
(theta, R) and is formed by the loader; seed 2, PCG64 — the pool `mc_tracing_inputs(oracle, 20_000, seed=2)` gave in rounds 1-5, bit for bit.
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

N, SEED = 20_000, 2


def main():
    from oracle import load_oracle
    from tools.mc_inputs import mc_elastic_matrices, mc_path_increment

    oracle = load_oracle()
    rng = np.random.Generator(np.random.PCG64(SEED))
    _, S = mc_elastic_matrices()
    tr = np.array([1.0, 1.0, 1.0, 0.0])
    theta = rng.uniform(-np.pi / 6 + 1e-5, np.pi / 6 - 1e-5, N)
    k_loads = rng.integers(0, 9, N)
    sn = np.zeros((N, 4))
    sn[:, :3] = 0.1
    for k in range(8):
        active = k_loads > k
        if not active.any():
            break
        d = mc_path_increment(theta[active], 0.7)
        _, s, *_ = oracle.mohr_coulomb(d @ S.T, sn[active], nthreads=8, tangent=False)
        dp = s @ tr / 3.0 - 0.1
        sn[active] = s - np.outer(dp, tr)          # :922-923
    R = rng.uniform(0.0, 0.7, N)
    assert not sn[:, 3].any()
    out = ROOT / "tests" / "golden" / "mc_tracing_pool.npz"
    np.savez_compressed(out, theta=theta, k_loads=k_loads.astype(np.int8), R=R, sigma_n3=np.ascontiguousarray(sn[:, :3]), seed=np.int64(SEED))
    print(f"wrote {out} ({out.stat().st_size} bytes): {N} points, {int((k_loads == 0).sum())} still hydrostatic")


if __name__ == "__main__":
    main()
