"""The C ABI used from plain C (examples/c_abi_demo.c): compiles against include/dxo.h with gcc, links libdxo_hip.so;
without a GPU it must refuse loudly (exit code 2, no CPU path), on an MI355X it must run the von Mises batch with host
arrays and with device-resident state and reproduce the history update."""
import pathlib
import shutil
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def demo(hip_library, tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    exe = tmp_path_factory.mktemp("cdemo") / "c_abi_demo"
    libdir = ROOT / "dolfinx_external_operator_amd"
    cmd = ["gcc", "-O2", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "c_abi_demo.c"), "-o", str(exe),
           f"-L{libdir}", "-ldxo_hip", f"-Wl,-rpath,{libdir}", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_links_and_refuses_without_a_device(demo):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([str(demo)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_runs_from_c_on_the_gpu(demo):
    r = subprocess.run([str(demo)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "history update reproduces the host result" in r.stdout
    assert "resident state: call and commit reproduce the plain call" in r.stdout
