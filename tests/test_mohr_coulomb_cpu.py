"""Mohr-Coulomb on the CPU (not gpu): the oracle against the reference-source goldens, known-answer
checks converted from the demo's printed/plotted verifications, and the per-lane device
math (csrc/mc_core.h, host build) against the oracle."""
import ctypes as C
import pathlib
import subprocess

import numpy as np
import pytest

from conftest import mc_compare as _compare
from conftest import mc_elastic_matrices, mc_path_increment, mc_tracing_inputs

ROOT = pathlib.Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden" / "mohr_coulomb.npz"



@pytest.fixture(scope="module")
def mc_core(tmp_path_factory):
    """Host build of the per-lane math (test aid only, never shipped)."""
    out = tmp_path_factory.mktemp("mc") / "mc_core_cpu.so"
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
                    f"-I{ROOT / 'dolfinx_external_operator_amd' / 'csrc'}", str(ROOT / "tests" / "helpers" / "mc_core_cpu.cpp"),
                    "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.mc_core_cpu.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8

    def run(deps, sn, **kw):
        from oracle.loader import mc_params

        prm = mc_params(**kw)
        deps = np.ascontiguousarray(deps, dtype=np.float64)
        sn = np.ascontiguousarray(sn, dtype=np.float64)
        n = len(deps)
        Ct, s = np.empty((n, 4, 4)), np.empty((n, 4))
        it, y, nr, dl = np.empty(n, dtype=np.int32), np.empty(n), np.empty(n), np.empty(n)
        lib.mc_core_cpu(C.byref(prm), n, *(a.ctypes.data for a in (deps, sn, Ct, s, it, y, nr, dl)))
        return Ct, s, it, y, nr, dl

    run.lib = lib
    return run


@pytest.mark.skipif(not GOLD.exists(), reason="golden not generated yet")
def test_oracle_matches_reference_source_golden(oracle):
    g = np.load(GOLD)
    prm = {k[4:]: g[k].item() for k in g.files if k.startswith("prm_")}
    prm["nitermax"] = int(prm["nitermax"])
    got = oracle.mohr_coulomb(g["deps"], g["sigma_n"], **prm)
    ref = (g["C_tang"], g["sigma"], g["niter"], g["yielding"], g["norm_res"], g["dlambda"])
    _compare(got, ref, "oracle vs reference golden", g["sigma_n"])
    # the golden covers elastic (1 iteration), plastic (2..) and the deps == 0 point (0 iterations, C_tang = 0)
    assert {0, 1, 2, 3} <= set(np.unique(g["niter"]).tolist())
    zero = np.flatnonzero(g["tag"] == -2)[0]
    assert g["niter"][zero] == 0 and np.all(g["C_tang"][zero] == 0.0)


@pytest.mark.skipif(not GOLD.exists(), reason="golden not generated yet")
def test_lane_math_matches_reference_source_golden(mc_core):
    g = np.load(GOLD)
    prm = {k[4:]: g[k].item() for k in g.files if k.startswith("prm_")}
    prm["nitermax"] = int(prm["nitermax"])
    got = mc_core(g["deps"], g["sigma_n"], **prm)
    ref = (g["C_tang"], g["sigma"], g["niter"], g["yielding"], g["norm_res"], g["dlambda"])
    _compare(got, ref, "mc_core vs reference golden", g["sigma_n"])


def test_lane_math_matches_oracle_on_tracing_distribution(oracle, mc_core):
    deps, sn = mc_tracing_inputs(oracle, 6000, seed=2)
    ref = oracle.mohr_coulomb(deps, sn, nthreads=8)
    got = mc_core(deps, sn)
    assert ref[2].max() <= 12 and (ref[3] > 0).mean() > 0.2      # converged mix of elastic and plastic points
    assert _compare(got, ref, "mc_core vs oracle (tracing)", sn) > 0.5


def test_lane_math_matches_oracle_with_shear_and_non_associated_flow(oracle, mc_core):
    deps, sn = mc_tracing_inputs(oracle, 3000, seed=5, shear=0.3)
    for kw in ({}, {"psi": 20 * np.pi / 180}, {"phi": 25 * np.pi / 180, "psi": 10 * np.pi / 180, "theta_T": 20 * np.pi / 180}):
        ref = oracle.mohr_coulomb(deps, sn, nthreads=8, **kw)
        got = mc_core(deps, sn, **kw)
        conv = ref[2] < 30
        assert conv.mean() > 0.95
        _compare(tuple(a[conv] for a in got), tuple(a[conv] for a in ref), f"mc_core vs oracle {kw}", sn[conv])


def test_elastic_points_return_c_elas_and_one_iteration(oracle):
    # demo_plasticity_mohr_coulomb.py:639-649: a small increment gives the elastic tangent
    Cel, S = mc_elastic_matrices()
    rng = np.random.default_rng(0)
    deps = rng.normal(0, 1e-6, (200, 4))
    sn = np.tile([-1.0, -1.2, -0.8, 0.1], (200, 1))
    Ct, s, it, y, nr, dl = oracle.mohr_coulomb(deps, sn)
    assert np.all(it == 1) and np.all(y < 0) and np.all(dl == 0)
    assert np.array_equal(Ct, np.broadcast_to(Cel, Ct.shape))
    assert np.allclose(s, sn + deps @ Cel.T, rtol=0, atol=1e-15)


def test_zero_increment_takes_zero_iterations(oracle):
    # deps == 0 -> norm_res0 == 0 -> 0/0 -> no iteration, C_tang = 0 (:500-505)
    Ct, s, it, y, nr, dl = oracle.mohr_coulomb(np.zeros((1, 4)), np.array([[0.1, 0.2, 0.1, 0.0]]))
    assert it[0] == 0 and np.all(Ct == 0.0) and np.array_equal(s[0], [0.1, 0.2, 0.1, 0.0])


def test_yield_surface_tracing_lands_on_the_surface(oracle):
    """Known-answer check converted from the demo's tracing block (:854-929): after the return the stress
    sits on f = 0 (the demo prints `max f`), elastic paths are untouched."""
    _, S = mc_elastic_matrices()
    n_angles = 50
    theta = np.linspace(-np.pi / 6 + 1e-5, np.pi / 6 - 1e-5, n_angles)
    tr = np.array([1.0, 1.0, 1.0, 0.0])
    sn = np.zeros((n_angles, 4))
    sn[:, :3] = 0.1
    saw_plastic = False
    for load in range(9):
        d = mc_path_increment(theta, 0.7)
        _, s, it, y, nr, dl = oracle.mohr_coulomb(d @ S.T, sn, tangent=False)
        f_after = oracle.mc_surface(s)[0]
        plastic = y > 0
        saw_plastic |= plastic.any()
        assert np.all(np.abs(f_after[plastic]) < 1e-7)
        assert np.all(it[~plastic] == 1) and np.allclose(s[~plastic], sn[~plastic] + d[~plastic])
        assert it.max() <= 6
        sn = s - np.outer(s @ tr / 3.0 - 0.1, tr)
    assert saw_plastic


def test_surface_is_continuous_across_the_abbo_sloan_transition(oracle):
    # K switches formula at |theta| = theta_T (:334-345); value and gradient are continuous there
    theta_T = 26 * np.pi / 180
    for sgn in (-1.0, 1.0):
        th = sgn * theta_T + np.array([-1e-7, 1e-7])
        rho, p = 2.0, -1.0
        s = np.stack([p + np.sqrt(2 / 3) * rho * np.cos(th_ - np.array([0, 2, 4])[k] * np.pi / 3 + np.pi / 6 * 0)
                      for th_ in th for k in range(3)]).reshape(2, 3)
        sig = np.zeros((2, 4))
        sig[:, :3] = s
        f, g, dg = oracle.mc_surface(sig)
        assert abs(f[1] - f[0]) < 1e-5 and np.max(np.abs(dg[1] - dg[0])) < 1e-4


def test_tangent_is_the_derivative_of_the_stress_map(oracle):
    """Kernel-level analogue of the demo's Taylor test (:1149-1235): sigma(deps + h v) - sigma(deps) - h C_tang v
    is second order in h on converged plastic points."""
    deps, sn = mc_tracing_inputs(oracle, 400, seed=9)
    Ct, s, it, y, nr, dl = oracle.mohr_coulomb(deps, sn)
    pl = (y > 0) & (it < 10)
    deps, sn, Ct, s = deps[pl][:100], sn[pl][:100], Ct[pl][:100], s[pl][:100]
    rng = np.random.default_rng(1)
    v = rng.normal(size=deps.shape)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    errs = []
    for h in (1e-6, 5e-7):
        s2 = oracle.mohr_coulomb(deps + h * v, sn, tangent=False)[1]
        errs.append(np.linalg.norm(s2 - s - h * np.einsum("nij,nj->ni", Ct, v), axis=1))
    rate = np.log2(np.median(errs[0]) / np.median(errs[1]))
    assert np.median(errs[0]) < 1e-5 * np.median(np.linalg.norm(h * np.einsum("nij,nj->ni", Ct, v), axis=1)) * 1e3
    assert rate > 1.5


def test_tangent_is_the_derivative_of_the_K_th_newton_iterate(oracle):
    """An AD-library-independent pin of what `jax.jacfwd(return_mapping)` means (demo_plasticity_mohr_coulomb.py:555):
    forward mode through `lax.while_loop` differentiates the ITERATES, so at a point that leaves the loop after K
    iterations C_tang = d sigma_K / d deps, where sigma_K is the K-th Newton iterate as a function of deps with the
    iteration count frozen. The frozen map is the same oracle run with tol = 0 and Nitermax = K (exactly K
    iterations, :503-522); its central finite difference must reproduce the oracle's tangent — and does NOT
    reproduce the implicit-function tangent of the converged point, from which the AD tangent differs by O(last
    Newton step). The stand-in AD backend of the golden generator is thereby cross-checked by
    plain differencing of the algorithm itself."""
    deps, sn = mc_tracing_inputs(oracle, 400, seed=11, shear=0.2)
    Ct, s, it, y, nr, dl = oracle.mohr_coulomb(deps, sn)
    pick = np.flatnonzero((y > 0) & (it >= 2) & (it <= 6))[:40]
    assert pick.size >= 20
    worst = 0.0
    for i in pick:
        K = int(it[i])
        fd = np.empty((4, 4))
        for j in range(4):
            h = 1e-6 * max(1e-3, abs(deps[i, j]))
            dp_, dm_ = deps[i].copy(), deps[i].copy()
            dp_[j] += h
            dm_[j] -= h
            sp = oracle.mohr_coulomb(dp_[None], sn[i][None], tol=0.0, nitermax=K, tangent=False)
            sm = oracle.mohr_coulomb(dm_[None], sn[i][None], tol=0.0, nitermax=K, tangent=False)
            assert sp[2][0] == K and sm[2][0] == K                      # the iteration count really is frozen
            fd[:, j] = (sp[1][0] - sm[1][0]) / (2 * h)
        err = np.max(np.abs(fd - Ct[i])) / np.max(np.abs(Ct[i]))
        worst = max(worst, err)
        assert err < 2e-8, (i, K, err)     # measured: max 8e-10 (differencing noise); the implicit-function tangent is off by up to 1.3e-7
    assert worst > 0.0


def test_near_the_meridians_the_lane_math_is_the_more_accurate_side(oracle, mc_core):
    """Why the tangent's parity bound is 1e-9 (not 1e-12) where 1 - |arg| < 1e-3 (conftest.mc_compare): there the REFERENCE's
    own derivative chain sin(3 asin(arg) / 3) cancels terms of size (1 - arg^2)^(-5/2) in fp64, while the lane math composes K
    with the Lode argument directly (csrc/mc_core.h). The referee is the reference's algorithm with 80-bit long double
    under the dual numbers (oracle_mohr_coulomb_ld): against it the fp64 restatement of the reference loses up to ~1e-10
    of the tangent's scale at the meridians, the lane math stays at rounding level — so the disagreement allowed there is
    the reference's rounding error, not the kernel's."""
    from conftest import lode_arg

    deps, sn = mc_tracing_inputs(oracle, 20000, seed=3)
    Co, so, ito, y, *_ = oracle.mohr_coulomb(deps, sn, nthreads=8)
    Cl, sl, itl = oracle.mohr_coulomb_long_double(deps, sn, nthreads=8)
    Ck, sk, itk, *_ = mc_core(deps, sn)
    plastic = y > 0
    sel = plastic & (ito == itl) & (itk == itl)                    # the tangent differentiates the iterates: compare equal counts only
    assert sel.sum() > 0.95 * plastic.sum()
    scale = np.max(np.abs(Cl))
    err_o = np.max(np.abs(Co - Cl).reshape(len(Co), -1), axis=1) / scale
    err_k = np.max(np.abs(Ck - Cl).reshape(len(Ck), -1), axis=1) / scale
    with np.errstate(all="ignore"):
        margin = np.minimum(1.0 - np.abs(lode_arg(sn)), 1.0 - np.abs(lode_arg(so)))
    margin = np.where(np.isfinite(margin), margin, 1.0)
    near = sel & (margin < 1e-3)
    far = sel & (margin >= 1e-3)
    assert near.sum() > 100 and far.sum() > 1000
    assert err_k[far].max() < 1e-12 and err_o[far].max() < 1e-12   # away from the meridians both sides are at rounding level
    assert err_k[near].max() < 1e-12                               # the lane math stays there at the meridians ...
    assert err_o[near].max() > 10 * err_k[near].max()              # ... the reference's fp64 chain does not
    assert np.max(np.abs(sk - sl)[sel]) < 1e-12 * max(np.max(np.abs(sl)), 1.0)


def test_dense_hessian_and_third_derivative_matrices_against_the_structured_operators(oracle, mc_core):
    """The pass forms hess(g) and T(t) = D_t hess(g) once as dense symmetric matrices (closed form of dev Q(s) dev for a
    deviatoric s) and applies them with 16 FMAs; the literal chain-rule operators they replaced stay in mc_core.h as the
    cross-check: same H v and T(t) v to rounding on stresses of the tracing distribution, both Abbo-Sloan branches."""
    lib = mc_core.lib
    lib.mc_dense_vs_structured.argtypes = [C.c_void_p] * 5
    deps, sn = mc_tracing_inputs(oracle, 400, seed=21, shear=0.3)
    from oracle.loader import mc_params

    prm = mc_params(psi=20 * np.pi / 180)     # non-associated: surf_eval<false> evaluates both surfaces
    rng = np.random.default_rng(3)
    worst = 0.0
    for i in range(deps.shape[0]):
        sig = np.ascontiguousarray(sn[i] + rng.normal(size=4) * 0.3)
        t, v = rng.normal(size=4), rng.normal(size=4)
        out = np.zeros(16)
        lib.mc_dense_vs_structured(C.byref(prm), sig.ctypes.data, t.ctypes.data, v.ctypes.data, out.ctypes.data)
        if not np.all(np.isfinite(out)):
            continue
        for a, b in ((out[0:4], out[4:8]), (out[8:12], out[12:16])):
            worst = max(worst, np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
    assert worst < 1e-11, worst      # the third-derivative terms cancel by several orders near the meridians


def test_lode_angle_sine_and_cosine_without_library_calls(mc_core):
    """sin(asin(u)/3) and cos(asin(u)/3) from Newton on the triple-angle cubics (mc_core.h lode_sin_cos) against libm over
    the whole range, including u -> 1 where asin is ill-conditioned and the branch switch at |u| = 1/2."""
    lib = mc_core.lib
    lib.mc_lode_sin_cos.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(0)
    u = np.concatenate([rng.uniform(-1, 1, 20000), 1 - np.logspace(-16, -2, 200), -(1 - np.logspace(-16, -2, 200)),
                        np.logspace(-300, -1, 200), [0.0, 0.5, -0.5, 0.5000000001, 1.0, -1.0]])
    sn, cs = np.zeros_like(u), np.zeros_like(u)
    lib.mc_lode_sin_cos(u.size, u.ctypes.data, sn.ctypes.data, cs.ctypes.data)
    th = np.arcsin(u) / 3
    # against a 40-digit evaluation: 3e-16 relative over (1e-38, 1], 1e-15 below the fp32 range of the seed (libm's chain: 3e-16)
    assert np.all(np.abs(sn - np.sin(th)) <= 2e-15 * np.abs(np.sin(th)))
    assert np.all(np.abs(cs - np.cos(th)) <= 1e-15)
    assert np.array_equal(np.signbit(sn), np.signbit(u))


def test_frozen_config4_pool_is_the_seeded_tracing_distribution(oracle):
    """tests/golden/mc_tracing_pool.npz (written by make_golden_mc_inputs.py) is what this suite's own seeded generator gives: the bench
    leg of config 4 and the 10^7-point test draw from the fixture, so that nothing under tools/ needs the checker to make inputs."""
    import ast
    import pathlib

    from tools import mc_inputs

    pool_d, pool_s = mc_inputs.mc_pool()
    deps, sn = mc_tracing_inputs(oracle, 20_000, seed=2)
    assert np.array_equal(pool_s, sn) and np.array_equal(pool_d, deps)
    d, s = mc_inputs.mc_pool_inputs(1000, seed=7)
    assert d.shape == (1000, 4) and s.shape == (1000, 4) and np.isfinite(d).all()
    yielding = oracle.mohr_coulomb(d, s, nthreads=2, tangent=False)[3]
    assert 0.1 < float((yielding > 0).mean()) < 0.9          # a mix of elastic and plastic points, as config 4 asks
    tree = ast.parse(pathlib.Path(mc_inputs.__file__).read_text())
    imported = {a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names} | \
               {n.module or "" for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)}
    assert not any(m.split(".")[0] == "oracle" for m in imported), imported
    assert "oracle" not in {n.id for n in ast.walk(tree) if isinstance(n, ast.Name)}
