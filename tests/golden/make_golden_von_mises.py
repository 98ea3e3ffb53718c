#!/usr/bin/env python3
"""Generate golden input/output vectors for the von Mises return-map path.

Run ONLY in the build container (needs /root/reference). The reference kernel is
not copied: this script parses `doc/demo/demo_plasticity_von_mises.py` with
`ast`, pulls out

  * the material-constant assignments   (demo_plasticity_von_mises.py:185-204)
  * the function `return_mapping`       (demo_plasticity_von_mises.py:298-332)

and executes them in a scratch namespace with `numba.njit` stubbed to the
identity and `PETSc.ScalarType = numpy.float64` (Numba/PETSc are absent here;
the stub only removes the JIT, the arithmetic is the reference's own NumPy).

d = 4 golden : the reference `return_mapping` itself, on (nc, nq=3, 4) arrays.
d = 6 golden : the reference's nested `_kernel` body (dimension agnostic apart
               from the literal 4 in the caller's allocation, :303) executed
               with the 3-D Mandel 6x6 `C_elas` / `deviatoric` built the same
               way the reference builds the 4x4 ones (:193-204).

history golden : the reference `return_mapping` driven through six load steps (loading, further loading, elastic
               unloading, reverse loading) with the demo's own state update between steps, `p += dp` and
               `sigma_n = sigma` (:564-565); every step's inputs and outputs are stored.

Outputs: tests/golden/von_mises_d4.npz, tests/golden/von_mises_d6.npz, tests/golden/von_mises_history_d4.npz
"""
import ast
import pathlib
import types

import numpy as np

REF = pathlib.Path("/root/reference/doc/demo/demo_plasticity_von_mises.py")
OUT = pathlib.Path(__file__).resolve().parent

CONST_NAMES = {"E", "nu", "E_tangent", "H", "sigma_0", "lmbda", "mu", "C_elas", "deviatoric"}


def _extract():
    tree = ast.parse(REF.read_text())
    picked = []
    kernel_src = None
    for node in tree.body:
        if isinstance(node, ast.Assign):
            names = set()
            for t in node.targets:
                for n in ast.walk(t):
                    if isinstance(n, ast.Name):
                        names.add(n.id)
            if names & CONST_NAMES:
                picked.append(node)
        elif isinstance(node, ast.AugAssign):
            base = node.target
            while isinstance(base, ast.Subscript):
                base = base.value
            if isinstance(base, ast.Name) and base.id in CONST_NAMES:
                picked.append(node)
        elif isinstance(node, ast.FunctionDef) and node.name == "return_mapping":
            picked.append(node)
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "_kernel":
                    kernel_src = ast.Module(body=[sub], type_ignores=[])
    assert kernel_src is not None
    return ast.Module(body=picked, type_ignores=[]), kernel_src


def _namespace():
    numba = types.SimpleNamespace(njit=lambda f: f)
    petsc = types.SimpleNamespace(ScalarType=np.float64)
    return {"np": np, "numba": numba, "PETSc": petsc}


def _inputs(rng, nc, nq, d, n_special):
    """Seeded inputs per SURVEY.md 8(d) + hand-picked special points at the front."""
    deps = rng.normal(0.0, 3e-3, size=(nc, nq, d))
    deps[..., 3:] *= np.sqrt(2.0)  # Mandel shear components
    sigma_n = rng.normal(0.0, 100.0, size=(nc, nq, d))
    p = np.abs(rng.normal(0.0, 1e-3, size=(nc, nq)))
    flat_deps = deps.reshape(-1, d)
    flat_sig = sigma_n.reshape(-1, d)
    flat_p = p.reshape(-1)
    # 0: tiny elastic increment from a stress-free state
    flat_deps[0] = 1e-7
    flat_sig[0] = 0.0
    flat_p[0] = 0.0
    # 1: deeply plastic uniaxial-ish increment
    flat_deps[1] = 0.0
    flat_deps[1, 0] = 5e-2
    flat_sig[1] = 0.0
    flat_p[1] = 0.0
    # 2: barely plastic (just above sigma_0 in pure shear)
    flat_deps[2] = 0.0
    flat_sig[2] = 0.0
    flat_sig[2, d - 1] = 250.0 * (1.0 + 1e-9) / np.sqrt(1.5)  # sigma_eq = sqrt(3/2) |s_shear|
    flat_p[2] = 0.0
    # 3: hardened state, elastic unloading
    flat_deps[3] = -1e-4
    flat_sig[3] = 0.0
    flat_sig[3, 0] = 300.0
    flat_p[3] = 0.1
    # 4: all-zero state: s == 0 exactly -> sigma_eq == 0 -> the reference yields NaN (0/0, :318-319)
    flat_deps[4] = 0.0
    flat_sig[4] = 0.0
    flat_p[4] = 0.0
    # 5: hydrostatic state (s == rounding noise), elastic
    flat_deps[5] = 0.0
    flat_deps[5, :3] = 1e-4
    flat_sig[5] = 0.0
    flat_sig[5, :3] = 10.0
    flat_p[5] = 0.0
    assert n_special >= 6
    return deps, sigma_n, p


def main():
    consts_and_rm, kernel_mod = _extract()

    # ---------------- d = 4: the reference function itself -----------------
    ns = _namespace()
    nc, nq, d = 200, 3, 4
    ns["num_quadrature_points"] = nq  # module global read by return_mapping (:295)
    exec(compile(consts_and_rm, str(REF), "exec"), ns)
    rng = np.random.Generator(np.random.PCG64(20240))
    deps, sigma_n, p = _inputs(rng, nc, nq, d, 6)
    with np.errstate(all="ignore"):
        C_tang, sigma, dp = ns["return_mapping"](deps, sigma_n, p)
    params = np.array([ns["E"], ns["nu"], ns["sigma_0"], ns["H"]], dtype=np.float64)
    np.savez(
        OUT / "von_mises_d4.npz",
        params=params, deps=deps, sigma_n=sigma_n, p=p,
        C_tang=C_tang, sigma=sigma, dp=dp,
        C_elas=ns["C_elas"], deviatoric=ns["deviatoric"],
    )
    plastic = np.count_nonzero(dp.reshape(-1) > 0)
    print(f"d=4: N={nc * nq} plastic={plastic} nan_points={np.count_nonzero(np.isnan(sigma).any(-1))}")

    # ---------------- d = 4 load history: return_mapping + the demo's state update (:564-565) --------------
    nc_h = 100
    rng = np.random.Generator(np.random.PCG64(777))
    base = rng.normal(0.0, 1.2e-3, size=(nc_h, nq, d))
    base[..., 3] *= np.sqrt(2.0)
    factors = [0.6, 1.0, 1.8, -0.4, -1.5, 0.9]          # per-step multiples of the base increment
    sig_hist = np.zeros((nc_h, nq, d))
    p_hist = np.zeros((nc_h, nq))
    steps = {}
    for k, fac in enumerate(factors):
        deps_k = base * fac
        with np.errstate(all="ignore"):
            C_k, s_k, dp_k = ns["return_mapping"](deps_k, sig_hist, p_hist)
        steps[f"deps_{k}"], steps[f"C_tang_{k}"], steps[f"sigma_{k}"], steps[f"dp_{k}"] = deps_k, C_k, s_k, dp_k
        p_hist = p_hist + dp_k.reshape(p_hist.shape)                    # p.x.petsc_vec.axpy(1.0, dp.x.petsc_vec), :564
        sig_hist = s_k.reshape(sig_hist.shape).copy()                   # sigma_n.x.array[:] = sigma...x.array, :565
        steps[f"p_after_{k}"], steps[f"sigma_n_after_{k}"] = p_hist.copy(), sig_hist.copy()
        print(f"history step {k}: factor {fac:+.1f} plastic {np.count_nonzero(dp_k > 0)}/{dp_k.size}")
    np.savez(OUT / "von_mises_history_d4.npz", params=params, n_steps=len(factors), **steps)

    # ---------------- d = 6: the reference _kernel body, 6x6 constants -----
    ns6 = _namespace()
    exec(compile(consts_and_rm, str(REF), "exec"), ns6)  # E, nu, H, sigma_0, lmbda, mu
    lmbda, mu = ns6["lmbda"], ns6["mu"]
    C6 = np.zeros((6, 6))
    C6[:3, :3] = lmbda
    C6[np.arange(6), np.arange(6)] += 2.0 * mu
    dev6 = np.eye(6)
    dev6[:3, :3] -= np.full((3, 3), 1.0 / 3.0)
    ns6["C_elas"] = C6
    ns6["deviatoric"] = dev6
    exec(compile(kernel_mod, str(REF), "exec"), ns6)
    kernel = ns6["_kernel"]
    nc, nq, d = 64, 8, 6
    rng = np.random.Generator(np.random.PCG64(20246))
    deps, sigma_n, p = _inputs(rng, nc, nq, d, 6)
    C_tang = np.empty((nc, nq, d, d))
    sigma = np.empty_like(sigma_n)
    dp = np.empty_like(p)
    with np.errstate(all="ignore"):
        for i in range(nc):
            for j in range(nq):
                C_tang[i, j], sigma[i, j], dp[i, j] = kernel(deps[i, j], sigma_n[i, j], p[i, j])
    np.savez(
        OUT / "von_mises_d6.npz",
        params=params, deps=deps, sigma_n=sigma_n, p=p,
        C_tang=C_tang, sigma=sigma, dp=dp, C_elas=C6, deviatoric=dev6,
    )
    plastic = np.count_nonzero(dp.reshape(-1) > 0)
    print(f"d=6: N={nc * nq} plastic={plastic} nan_points={np.count_nonzero(np.isnan(sigma).any(-1))}")


if __name__ == "__main__":
    main()
