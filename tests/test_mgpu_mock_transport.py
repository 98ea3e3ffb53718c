"""The N > 1 data path of libdxo's multi-GPU entry point on ONE GPU, over a mock transport (gpu).

RCCL refuses two ranks on one device and the builder's box has one MI355X, so `dxo_mgpu_von_mises` with more than one rank had
never moved a byte. tests/mock_rccl/mock_rccl.cpp stands in for librccl.so.1 (the library resolves RCCL with dlopen / dlsym, so
a directory in front of LD_LIBRARY_PATH is enough — in a process that has not loaded the real one, hence the plain C++ driver
and no torch): `world` ranks of one process share device 0, collectives and send / receive pairs become device-to-device copies
at ncclGroupEnd. What is checked is WHICH bytes land WHERE: for every gather form (all-gather of everything, compact, compact
as direct send / receive pairs, compact in 1 / 4 / 7 overlapped pieces) the full-length (C_tang, sigma, dp) of EVERY rank must
equal, bit for bit, the blocks computed one at a time by a world-of-one group, and the transport must have carried exactly
world x (world - 1) x n x bytes-per-point bytes; the Mohr-Coulomb entry point (five outputs of three element sizes, one not
requested) goes through the same split and all-gather. The reference has no counterpart (it never gathers:
src/dolfinx_external_operator/external_operator.py:365-371, 445); north_star's design asks for the exchange.
"""
import os
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
HERE = ROOT / "tests" / "mock_rccl"


@pytest.fixture(scope="module")
def driver(hip_library, tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not pathlib.Path(hipcc).exists():
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("mock_rccl")
    libdir = ROOT / "dolfinx_external_operator_amd"
    r = subprocess.run([hipcc, "-O2", "-fPIC", "-shared", str(HERE / "mock_rccl.cpp"), "-o", str(out / "librccl.so.1")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([hipcc, "-O2", f"-I{ROOT / 'include'}", str(HERE / "mgpu_world_test.cpp"), "-o", str(out / "mgpu_world_test"),
                        f"-L{libdir}", "-ldxo_hip", f"-L{out}", "-l:librccl.so.1", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{out}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def test_mock_transport_builds_without_a_gpu(driver):
    """CPU check: the stand-in and the driver compile and link against the C ABI (nothing runs)."""
    assert (driver / "librccl.so.1").exists() and (driver / "mgpu_world_test").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("world, n", [(2, 10_000), (3, 10_000), (4, 6_400), (8, 1_280)])
def test_every_gather_form_leaves_the_block_by_block_result_on_every_rank(driver, world, n):
    env = dict(os.environ, LD_LIBRARY_PATH=f"{driver}:{os.environ.get('LD_LIBRARY_PATH', '')}")
    r = subprocess.run([str(driver / "mgpu_world_test"), str(world), str(n)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("ok", "MISMATCH"))]
    assert len(lines) == 7 and all(ln.startswith("ok") for ln in lines), r.stdout      # six von Mises forms + the Mohr-Coulomb full gather
