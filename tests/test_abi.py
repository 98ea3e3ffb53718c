"""The C-ABI library: builds for gfx950, loads, exports what include/dxo.h declares (not gpu)."""
import ctypes as C
import pathlib
import re
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "dxo.h"


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(dxo_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_path():
    names = header_functions()
    for must in ("dxo_ctx_create", "dxo_ctx_destroy", "dxo_von_mises", "dxo_heat", "dxo_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(hip_library):
    for name in header_functions():
        assert hasattr(hip_library, name), f"{name} declared in include/dxo.h but not exported"


def test_binding_covers_every_declared_symbol(hip_library):
    from dolfinx_external_operator_amd._lib import declared_symbols

    assert sorted(declared_symbols()) == header_functions()


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "dxo.h"\nint main(void){ dxo_vm_params p = {70e3, 0.3, 250.0, 707.07}; (void)p; return DXO_ABI_VERSION - 2; }\n')
    res = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{ROOT / 'include'}", "-c", str(src), "-o",
                          str(tmp_path / "t.o")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_code_object_targets_gfx950(hip_library):
    from dolfinx_external_operator_amd._lib import LIB_PATH

    blob = LIB_PATH.read_bytes()
    assert b"gfx950" in blob
    assert b"vm_tile" in blob and b"heat_g2" in blob


def test_argument_errors_without_a_device(hip_library):
    lib = hip_library
    assert lib.dxo_abi_version() == 2          # 2: dxo_assign_desc::elem_bytes (float32 / float64 / complex128 through the device assigner)
    assert lib.dxo_ctx_create(0, None) == -1                       # DXO_E_NULL
    assert lib.dxo_von_mises(None, None, 4, 0, 0, None, None, None, None, None, None) == -1
    assert lib.dxo_heat(None, 1.0, 1.0, 2, 0, 0, None, None, None, None, None) == -1
    assert lib.dxo_ctx_destroy(None) == -1
    h = C.c_void_p()
    assert lib.dxo_vm_state_create(None, 6, 10, C.byref(h)) == -1  # dxo_vm_state_*: a NULL ctx is refused before any HIP call
    assert lib.dxo_vm_state_upload(None, None, 0, None, None) == -1
    assert lib.dxo_vm_state_commit(None, None) == -1
    assert lib.dxo_von_mises_state(None, None, None, 0, None, None, None, None) == -1
    assert lib.dxo_von_mises_field_state(None, None, None, None, 0, None, None, None, None) == -1
    lib.dxo_vm_state_destroy(None, None)                           # no-op
    n = C.c_int(-5)
    rc = lib.dxo_device_count(C.byref(n))
    assert rc in (0, -7) and n.value >= 0


def test_mgpu_entry_points_reject_bad_arguments_without_a_device(hip_library):
    """dxo_mgpu_*: exported, typed, and argument errors come back before RCCL or a GPU is touched."""
    lib = hip_library
    h = C.c_void_p()
    assert lib.dxo_mgpu_create(None, 1, None) == -1                # DXO_E_NULL
    assert lib.dxo_mgpu_create(None, 0, C.byref(h)) == -3          # DXO_E_SIZE
    assert lib.dxo_mgpu_destroy(None) == -1
    assert lib.dxo_mgpu_create_local(None, 1, None) == -1
    assert lib.dxo_mgpu_create_local(None, 0, C.byref(h)) == -3
    assert lib.dxo_mgpu_von_mises_host(None, None, 6, 0, None, None, None, None, None, None) == -1
    assert lib.dxo_mgpu_size(None) == -1
    assert lib.dxo_mgpu_unique_id(None) == -1
    assert lib.dxo_mgpu_create_rank(None, None, 0, 1, C.byref(h)) == -1
    assert lib.dxo_mgpu_ctx(None, 0) is None
    n = C.c_int(0)
    if lib.dxo_device_count(C.byref(n)) != 0 or n.value == 0:
        assert lib.dxo_mgpu_create(None, 2, C.byref(h)) == -7      # DXO_E_NODEVICE: no silent CPU path
        assert not h.value


def test_libdxo_does_not_link_rccl():
    """RCCL is resolved lazily (dlopen at the first dxo_mgpu_* call): single-GPU users never load the 570 MB library."""
    from dolfinx_external_operator_amd._lib import LIB_PATH

    res = subprocess.run(["readelf", "-d", str(LIB_PATH)], capture_output=True, text=True)
    assert res.returncode == 0
    assert "librccl" not in res.stdout and "libamdhip64" in res.stdout


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "dolfinx_external_operator_amd"
    for f in pkg.rglob("*.py"):
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
        assert "libdxo_oracle" not in text, f
    for f in (pkg / "csrc").iterdir():
        text = f.read_text()
        assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', text), f   # no oracle source is compiled in
        assert "dxo_oracle" not in text and "oracle_von_mises" not in text and "oracle_mohr" not in text, f


def test_missing_library_is_a_loud_error(tmp_path):
    from dolfinx_external_operator_amd._lib import DxoError, load_library

    with pytest.raises(DxoError, match="no CPU fallback"):
        load_library(tmp_path / "libdxo_hip.so")
