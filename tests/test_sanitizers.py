"""AddressSanitizer + UBSan over the CPU-side code that shares its math with the kernels (not gpu).
GPU sanitizers are unavailable on the pool; this covers the oracle and the host build of csrc/mc_core.h."""
import pathlib
import subprocess

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_oracle_and_lane_math_are_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "sanitize_main"
    obj = tmp_path / "dxo_oracle.o"
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fopenmp"]
    subprocess.run(["gcc", "-std=c11", *flags, "-c", str(ROOT / "oracle" / "dxo_oracle.c"), "-o", str(obj)], check=True)
    subprocess.run(["g++", "-std=c++17", *flags, f"-I{ROOT / 'dolfinx_external_operator_amd' / 'csrc'}",
                    str(ROOT / "tests" / "helpers" / "sanitize_main.cpp"), str(ROOT / "oracle" / "mc_oracle.cpp"), str(obj),
                    "-o", str(exe), "-lm"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env={"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1", "OMP_NUM_THREADS": "2"})
    assert res.returncode == 0, res.stdout + res.stderr
    assert "runtime error" not in res.stderr and "ERROR: AddressSanitizer" not in res.stderr, res.stderr
    assert "sanitize harness: ok" in res.stdout


def test_host_half_is_clean_under_tsan_and_asan(tmp_path):
    """The product's host-side code — csrc/host_pool.h (worker threads) and csrc/vm_host.h (tangent rebuild from
    (sigma, dp)) — has no HIP dependency and is built on its own under ThreadSanitizer and under ASan + UBSan."""
    inc = f"-I{ROOT / 'dolfinx_external_operator_amd' / 'csrc'}"
    src = str(ROOT / "tests" / "helpers" / "host_half.cpp")
    for name, san in (("tsan", ["-fsanitize=thread"]), ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])):
        exe = tmp_path / f"host_half_{name}"
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-DHOST_HALF_MAIN", *san, "-fno-omit-frame-pointer", inc, src, "-o", str(exe),
                        "-lpthread"], check=True)
        res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout + res.stderr
        assert "WARNING: ThreadSanitizer" not in res.stderr and "ERROR: AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr
        assert "host half harness: ok" in res.stdout
