import os
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference kernels (oracle/, test infrastructure only)."""
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def hip_library():
    """libdxo_hip.so, built in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    from dolfinx_external_operator_amd._build import build_library
    from dolfinx_external_operator_amd._lib import load_library

    build_library()
    return load_library()


@pytest.fixture(scope="session")
def ctx(hip_library):
    """A dxo_ctx on GPU 0. No skip: on the GPU box a missing device/library must fail loudly."""
    from dolfinx_external_operator_amd import Context

    c = Context(0)
    # one context serves the whole session: a calibrated block that one test frees must not be handed to the next test's request of the same size
    # (tests read the calibration's own record); the retained block has a test of its own, which switches it on (tests/test_round2_gpu.py)
    c.set_option("placement_cache", 0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def experiments_build(ctx):
    """True when the library under test was built with -DDXO_EXPERIMENTS (scripts/exp/build_variant.py + DXO_HIP_LIBRARY): the kernel
    variants that were measured and not shipped (ICNN pipelined / hybrid, the patch form of the internal force) exist only there. The
    product build must REFUSE their options."""
    try:
        ctx.set_option("adjoint_patch", 1)
    except ValueError:      # DXO_E_OPTION (argument errors are ValueError in the binding)
        return False
    ctx.set_option("adjoint_patch", 0)
    return True


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(GOLDEN / name)

    return _load


def vm_inputs(n, d, seed, plastic_scale=1.0):
    """Seeded von Mises inputs, SURVEY.md 8(d): deps~N(0,3e-3) (Mandel shear x sqrt2), sigma_n~N(0,100), p=|N(0,1e-3)|."""
    rng = np.random.Generator(np.random.PCG64(seed))
    deps = rng.normal(0.0, 3e-3 * plastic_scale, size=(n, d))
    deps[:, 3:] *= np.sqrt(2.0)
    sigma_n = rng.normal(0.0, 100.0 * plastic_scale, size=(n, d))
    p = np.abs(rng.normal(0.0, 1e-3, size=n))
    return deps, sigma_n, p


def vm_indeterminate_sigma0(evaluate, d, a=384.0, span=16):
    """A yield stress that puts the uniaxial state sigma_n = (a, 0, ...), deps = 0, p = 0 EXACTLY on the yield surface in
    the arithmetic of `evaluate`: f_elastic = sigma_eq - sigma_0 == 0, where the reference's n_elas is 0/0 and its
    tangent all NaN (demo_plasticity_von_mises.py:318). sigma_eq is |a| up to a few roundings that depend on the
    operation order (dense mat-vecs in the oracle, sparse + FMA in the kernel), so the neighbours of |a| are tried.
    `evaluate(deps, sigma_n, p, sigma_0) -> (C_tang, sigma, dp)` on (1, d) inputs. Returns (sigma_0, sigma_n_row)."""
    sn = np.zeros((1, d))
    sn[0, 0] = a
    cand = [a]
    lo = hi = a
    for _ in range(span):
        lo, hi = np.nextafter(lo, -np.inf), np.nextafter(hi, np.inf)
        cand += [lo, hi]
    for s0 in cand:
        with np.errstate(all="ignore"):
            C, s, dp = evaluate(np.zeros((1, d)), sn, np.zeros(1), float(s0))
        C, s, dp = np.asarray(C).reshape(-1), np.asarray(s).reshape(-1), np.asarray(dp).reshape(-1)
        if np.isnan(C).all() and np.isfinite(s).all() and dp[0] == 0.0:
            return float(s0), sn[0].copy()
    raise AssertionError("no yield stress within the searched neighbourhood makes f_elastic == 0 exactly")


def assert_close_scaled(actual, expected, rtol, what=""):
    """max |a-b| <= rtol * max|b| over finite entries, and identical NaN pattern."""
    actual = np.asarray(actual).reshape(-1)
    expected = np.asarray(expected).reshape(-1)
    assert actual.shape == expected.shape, f"{what}: shape {actual.shape} vs {expected.shape}"
    nan_a, nan_e = np.isnan(actual), np.isnan(expected)
    assert np.array_equal(nan_a, nan_e), f"{what}: NaN pattern differs ({nan_a.sum()} vs {nan_e.sum()})"
    m = ~nan_e
    if not m.any():
        return
    scale = np.max(np.abs(expected[m]))
    err = np.max(np.abs(actual[m] - expected[m]))
    assert err <= rtol * max(scale, np.finfo(float).tiny), f"{what}: err {err:.3e} > {rtol:.1e} * {scale:.3e}"


# ------------------------------------------------------------------------------- Mohr-Coulomb inputs
# constants and the path increment live in tools/mc_inputs.py (bench legs draw from the frozen pool there, without the checker); the seeded
# generator below is the tests' own and DOES use the checker to advance the states along the tracing loads
from tools.mc_inputs import MC_E, MC_NU, mc_elastic_matrices, mc_path_increment  # noqa: E402,F401


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


# C_tang entries are O(E) = 8e3: agreement is asked relative to the tangent's scale (see mc_compare).
RTOL_S = 1e-12


def lode_arg(sig):
    """arg of theta() (:290-293) for stresses (n,4)."""
    dev = sig.copy()
    dev[:, :3] -= dev[:, :3].mean(axis=1, keepdims=True)
    J2 = 0.5 * np.sum(dev * dev, axis=1)
    J3 = dev[:, 2] * (dev[:, 0] * dev[:, 1] - dev[:, 3] ** 2 / 2.0)
    with np.errstate(all="ignore"):
        return -(3.0 * np.sqrt(3.0) * J3) / (2.0 * np.sqrt(J2 ** 3))


def mc_compare(got, ref, what, sigma_n):
    """Parity of (C_tang, sigma, niter, yielding, norm_res, dlambda), fp64.

    C_tang: |err| <= 1e-9 * max|C_tang| on every point, and <= 1e-12 * max|C_tang| on points whose start and
    end states keep margin = 1 - |arg| > 1e-3 (arg = the asin argument of the Lode angle, :292). The looser
    bound only matters at the compression / extension meridians where the demo's tracing paths end
    (margin -> 0): there the REFERENCE's own derivative chain sin(3 asin(arg)/3) cancels terms of size
    (1 - arg^2)^(-5/2); the HIP lane math uses sin(3 theta) = arg instead (csrc/mc_core.h) and is the more
    accurate side. Measured between oracle and lane math: 2e-15 at margin > 0.1, 5e-13 at 1e-5, 3e-10 below
    1e-7. sigma 1e-12, yielding 1e-12, dlambda 1e-12 (absolute), norm_res 1e-10 (absolute; it is rounding
    noise at convergence), iteration counts exact. Returns the fraction of points held to 1e-12."""
    Cg, sg, itg, yg, nrg, dlg = got
    Cr, sr, itr, yr, nrr, dlr = ref
    assert np.array_equal(itg, itr), f"{what}: iteration counts differ at {np.flatnonzero(itg != itr)[:10]}"
    scale_C = np.max(np.abs(Cr))
    with np.errstate(all="ignore"):
        margin = np.minimum(1.0 - np.abs(lode_arg(np.asarray(sigma_n, dtype=float).reshape(-1, 4))), 1.0 - np.abs(lode_arg(sr)))
        margin = np.where(np.isfinite(margin) & (yr > 0), margin, 1.0)  # elastic / J2 = 0: theta's derivatives are not involved
    tol = np.where(margin > 1e-3, 1e-12, 1e-9)
    errC = np.max(np.abs(Cg - Cr).reshape(len(Cr), -1), axis=1) / scale_C
    bad = errC > tol
    assert not bad.any(), (f"{what}: C_tang rel err {errC[bad].max():.3e} at margin {margin[bad][np.argmax(errC[bad])]:.2e} "
                           f"(allowed {tol[bad][np.argmax(errC[bad])]:.0e}); {bad.sum()} points")
    assert np.max(np.abs(sg - sr)) <= RTOL_S * max(np.max(np.abs(sr)), 1.0), f"{what}: sigma"
    finy = np.isfinite(yr)   # a hydrostatic trial stress has J2 = 0 and f = NaN in the reference too
    assert np.array_equal(finy, np.isfinite(yg)), f"{what}: yielding NaN pattern"
    assert np.max(np.abs(yg[finy] - yr[finy]), initial=0.0) <= 1e-12 * max(np.max(np.abs(yr[finy]), initial=1.0), 1.0), f"{what}: yielding"
    assert np.max(np.abs(dlg - dlr)) <= 1e-12, f"{what}: dlambda"
    fin = np.isfinite(nrr)
    assert np.array_equal(fin, np.isfinite(nrg))
    assert np.max(np.abs(nrg[fin] - nrr[fin]), initial=0.0) <= 1e-10, f"{what}: norm_res"
    return float((tol <= 1e-12).mean())


def mc_compare_all(got, ref, what, sigma_n, nitermax=200, slow=30):
    """mc_compare on the points the oracle converges in fewer than `slow` iterations (iteration counts exact,
    tangent 1e-12 / 1e-9, see mc_compare) PLUS the rest, which round 1 filtered out:
      * the sets of points that hit Nitermax are equal (non-convergence is reported identically, :469, :533);
      * slowly converging points (slow <= niter < Nitermax): both sides converged, so both returned stresses solve the
        same equations to the Newton tolerance: sigma agrees to 1e-7 of its scale, the iteration count to +-2 (a
        residual sitting at the tolerance may need one more step on one side), norm_res <= tol on both;
      * points at Nitermax: the iterates wander (no fixed point was found), only finiteness / NaN pattern of sigma is
        compared.
    Returns (fraction fast, number slow, number non-converged)."""
    itg, itr = got[2], ref[2]
    fast = itr < slow
    frac = mc_compare(tuple(a[fast] for a in got), tuple(a[fast] for a in ref), what + " [fast points]",
                      np.asarray(sigma_n).reshape(-1, 4)[fast])
    dead_r, dead_g = itr >= nitermax, itg >= nitermax
    assert np.array_equal(dead_r, dead_g), f"{what}: points at Nitermax differ: {np.flatnonzero(dead_r != dead_g)[:10]}"
    slowm = ~fast & ~dead_r
    if slowm.any():
        assert np.max(np.abs(itg[slowm].astype(int) - itr[slowm].astype(int))) <= 2, f"{what}: slow points' iteration counts"
        scale = max(np.max(np.abs(ref[1][slowm])), 1.0)
        assert np.max(np.abs(got[1][slowm] - ref[1][slowm])) <= 1e-7 * scale, f"{what}: sigma on slowly converging points"
        assert np.max(got[4][slowm]) <= 1e-8 and np.max(ref[4][slowm]) <= 1e-8, f"{what}: norm_res on slowly converging points"
    if dead_r.any():
        assert np.array_equal(np.isfinite(got[1][dead_r]), np.isfinite(ref[1][dead_r])), f"{what}: sigma NaN pattern at Nitermax"
    return frac, int(slowm.sum()), int(dead_r.sum())
