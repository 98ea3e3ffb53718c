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
    yield c
    c.close()


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
