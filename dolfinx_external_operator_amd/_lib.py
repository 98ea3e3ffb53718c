"""ctypes binding of libdxo_hip.so (the C ABI declared in include/dxo.h).

There is deliberately no CPU fallback: if the HIP library cannot be loaded, or no GPU context
can be created, the operators raise.
"""
from __future__ import annotations

import collections
import ctypes as C
import pathlib
import threading
import weakref

import numpy as np

PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = PKG / "libdxo_hip.so"

MEM_HOST = 0
MEM_DEVICE = 1

ERRORS = {
    -1: "DXO_E_NULL", -2: "DXO_E_DIM", -3: "DXO_E_SIZE", -4: "DXO_E_MEM",
    -5: "DXO_E_ALIGN", -6: "DXO_E_OPTION", -7: "DXO_E_NODEVICE",
}


class VmParams(C.Structure):
    """dxo_vm_params — demo_plasticity_von_mises.py:185-188."""
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("sigma_0", C.c_double), ("H", C.c_double)]


class McParams(C.Structure):
    """dxo_mc_params — demo_plasticity_mohr_coulomb.py:110-116, 469."""
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("c", C.c_double), ("phi", C.c_double),
                ("psi", C.c_double), ("theta_T", C.c_double), ("a", C.c_double), ("tol", C.c_double),
                ("nitermax", C.c_int32), ("_pad", C.c_int32)]


class IsiharaParams(C.Structure):
    """dxo_isihara_params — W = c1 (I1bar-3) + c2 (I2bar-3) + c3 (I1bar-3)^2 + c4 (J-1)^2 (demo_hyperelasticity.py:700)."""
    _fields_ = [("c1", C.c_double), ("c2", C.c_double), ("c3", C.c_double), ("c4", C.c_double)]


ABI_VERSION = 2      # include/dxo.h DXO_ABI_VERSION (2: dxo_assign_desc::elem_bytes)


class AssignDesc(C.Structure):
    """dxo_assign_desc — one subspace of a dofmap assigner (external_operator.py:286-335)."""
    _fields_ = [("n_cells", C.c_int64), ("n_pts", C.c_int32), ("val_size", C.c_int32), ("offset", C.c_int32),
                ("n_points_total", C.c_int32), ("comp_size", C.c_int32), ("elem_bytes", C.c_int32)]


class IcnnWeights(C.Structure):
    """dxo_icnn_weights — the reference's state_dict tensors (demo_hyperelasticity.py:302-315), fp32."""
    _fields_ = [(k, C.c_void_p) for k in ("layers0_weight", "layers0_bias", "layers1_weights", "skip1_weight", "skip1_bias",
                                            "layers2_weights", "skip2_weight", "skip2_bias", "layers3_weights",
                                            "skip3_weights")] + [("n_hidden", C.c_int32), ("_pad", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double),
                ("total_ms", C.c_double)]


PLACEMENT_MAX = 32


class PlacementInfo(C.Structure):
    """dxo_placement_info — what dxo_output_alloc's calibration saw."""
    _fields_ = [("mode", C.c_int16), ("probe_kind", C.c_int16), ("candidates", C.c_int32), ("chosen", C.c_int32), ("vmm_mask", C.c_uint32),
                ("probe_GBps", C.c_double * PLACEMENT_MAX), ("calibration_ms", C.c_double), ("chosen_GBps", C.c_double),
                ("tuned_blocks_per_cu", C.c_int32), ("rounds", C.c_int32)]


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("total_mem_bytes", C.c_int64)]


_P = C.c_void_p
_SIGNATURES = {
    "dxo_abi_version": (C.c_int, []),
    "dxo_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dxo_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "dxo_ctx_destroy": (C.c_int, [_P]),
    "dxo_last_error": (C.c_char_p, [_P]),
    "dxo_ctx_device_info": (C.c_int, [_P, C.POINTER(DeviceInfo)]),
    "dxo_ctx_set_stream": (C.c_int, [_P, _P]),
    "dxo_ctx_synchronize": (C.c_int, [_P]),
    "dxo_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "dxo_ctx_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "dxo_last_timing": (C.c_int, [_P, C.POINTER(Timing)]),
    "dxo_host_alloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "dxo_host_free": (C.c_int, [_P, _P]),
    "dxo_host_register": (C.c_int, [_P, _P, C.c_int64]),
    "dxo_host_unregister": (C.c_int, [_P, _P]),
    "dxo_output_alloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "dxo_output_free": (C.c_int, [_P, _P]),
    "dxo_output_info": (C.c_int, [_P, _P, C.POINTER(PlacementInfo)]),
    "dxo_output_alloc_probed": (C.c_int, [_P, C.c_int64, _P, _P, C.c_double, C.POINTER(C.c_int32), C.c_int, C.POINTER(_P)]),
    "dxo_vm_output_alloc": (C.c_int, [_P, C.c_int, C.c_int64, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "dxo_von_mises": (C.c_int, [_P, C.POINTER(VmParams), C.c_int, C.c_int64, C.c_int] + [_P] * 6),
    "dxo_vm_expand_tangent": (C.c_int, [_P, C.POINTER(VmParams), C.c_int, C.c_int64, C.c_int, _P, _P, _P]),
    "dxo_vm_clear_marks": (C.c_int, [_P, C.c_int64, _P]),
    "dxo_vm_commit_state": (C.c_int, [_P, C.c_int, C.c_int64, _P, _P, _P, _P]),
    "dxo_vm_state_create": (C.c_int, [_P, C.c_int, C.c_int64, C.POINTER(_P)]),
    "dxo_vm_state_destroy": (None, [_P, _P]),
    "dxo_vm_state_upload": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "dxo_vm_state_download": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "dxo_vm_state_commit": (C.c_int, [_P, _P]),
    "dxo_vm_state_pointers": (C.c_int, [_P, _P] + [C.POINTER(_P)] * 4),
    "dxo_von_mises_state": (C.c_int, [_P, C.POINTER(VmParams), _P, C.c_int, _P, _P, _P, _P]),
    "dxo_von_mises_field_state": (C.c_int, [_P, C.POINTER(VmParams), _P, _P, C.c_int, _P, _P, _P, _P]),
    "dxo_device_alloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "dxo_device_free": (C.c_int, [_P, _P]),
    "dxo_copy": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int]),
    "dxo_heat": (C.c_int, [_P, C.c_double, C.c_double, C.c_int, C.c_int64, C.c_int] + [_P] * 5),
    "dxo_mohr_coulomb": (C.c_int, [_P, C.POINTER(McParams), C.c_int64, C.c_int] + [_P] * 8),
    "dxo_mc_state_create": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "dxo_mc_state_destroy": (None, [_P, _P]),
    "dxo_mc_state_upload": (C.c_int, [_P, _P, C.c_int, _P]),
    "dxo_mc_state_download": (C.c_int, [_P, _P, C.c_int, _P]),
    "dxo_mc_state_commit": (C.c_int, [_P, _P]),
    "dxo_mc_state_pointers": (C.c_int, [_P, _P, C.POINTER(_P), C.POINTER(_P)]),
    "dxo_mohr_coulomb_state": (C.c_int, [_P, C.POINTER(McParams), _P, C.c_int] + [_P] * 7),
    "dxo_mohr_coulomb_field": (C.c_int, [_P, C.POINTER(McParams), _P, C.c_int] + [_P] * 8),
    "dxo_icnn_field": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, _P, _P]),
    "dxo_isihara_field": (C.c_int, [_P, C.POINTER(IsiharaParams), _P, C.c_int, _P, _P, _P]),
    "dxo_conductivity": (C.c_int, [_P, C.c_double, C.c_double, C.c_int64, C.c_int, _P, _P, _P]),
    "dxo_mc_summary": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_int, _P, _P, _P, _P]),
    "dxo_icnn_create": (C.c_int, [_P, C.POINTER(IcnnWeights), C.POINTER(_P)]),
    "dxo_icnn_destroy": (C.c_int, [_P, _P]),
    "dxo_icnn_correction": (C.c_int, [_P, _P, _P]),
    "dxo_icnn_eval": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, _P, _P, _P]),
    "dxo_mesh_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "dxo_mesh_destroy": (C.c_int, [_P, _P]),
    "dxo_operand_value_size": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "dxo_eval_operand": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "dxo_mesh_set_facet_tables": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "dxo_eval_operand_facets": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "dxo_von_mises_field": (C.c_int, [_P, C.POINTER(VmParams), _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "dxo_assign": (C.c_int, [_P, C.POINTER(AssignDesc), _P, _P, _P, C.c_int64]),
    "dxo_assign_plan_create": (C.c_int, [_P, C.POINTER(AssignDesc), _P, C.c_int64, C.POINTER(_P)]),
    "dxo_assign_plan_destroy": (None, [_P, _P]),
    "dxo_assign_apply": (C.c_int, [_P, _P, _P, _P]),
    "dxo_assign_plan_form": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dxo_mesh_set_weights": (C.c_int, [_P, _P, _P]),
    "dxo_mesh_set_coordinate_values": (C.c_int, [_P, _P, _P]),
    "dxo_eval_coordinate": (C.c_int, [_P, _P, C.c_int, _P, C.c_int64, _P]),
    "dxo_operand_adjoint": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int64, _P]),
    "dxo_tangent_apply": (C.c_int, [_P, _P, _P, _P, _P]),
    "dxo_tangent_diagonal": (C.c_int, [_P, _P, _P, _P]),
    "dxo_tangent_apply_vm": (C.c_int, [_P, _P, C.POINTER(VmParams), _P, _P, _P, _P]),
    "dxo_von_mises_residual": (C.c_int, [_P, C.POINTER(VmParams), _P, _P, _P, _P, _P, _P, _P]),
    "dxo_tangent_diagonal_vm": (C.c_int, [_P, _P, C.POINTER(VmParams), _P, _P, _P]),
    "dxo_heat_field": (C.c_int, [_P, C.c_double, C.c_double, _P, C.c_int, _P, _P, _P, _P]),
    "dxo_isihara": (C.c_int, [_P, C.POINTER(IsiharaParams), C.c_int64, C.c_int, _P, _P, _P]),
    "dxo_mgpu_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "dxo_mgpu_create_local": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    "dxo_mgpu_von_mises_host": (C.c_int, [_P, C.POINTER(VmParams), C.c_int, C.c_int64] + [_P] * 6),
    "dxo_mgpu_unique_id": (C.c_int, [_P]),
    "dxo_mgpu_create_rank": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "dxo_mgpu_destroy": (C.c_int, [_P]),
    "dxo_mgpu_size": (C.c_int, [_P]),
    "dxo_mgpu_local_count": (C.c_int, [_P]),
    "dxo_mgpu_rank": (C.c_int, [_P, C.c_int]),
    "dxo_mgpu_ctx": (_P, [_P, C.c_int]),
    "dxo_mgpu_last_error": (C.c_char_p, [_P]),
    "dxo_mgpu_synchronize": (C.c_int, [_P]),
    "dxo_mgpu_all_gather": (C.c_int, [_P, C.POINTER(_P), C.c_int64]),
    "dxo_mgpu_von_mises": (C.c_int, [_P, C.POINTER(VmParams), C.c_int, C.c_int64, C.c_int] + [C.POINTER(_P)] * 6),
    "dxo_mgpu_mohr_coulomb": (C.c_int, [_P, C.POINTER(McParams), C.c_int64, C.c_int] + [C.POINTER(_P)] * 8),
    "dxo_mgpu_icnn": (C.c_int, [_P, C.POINTER(_P), C.c_int, C.c_int64, C.c_int] + [C.POINTER(_P)] * 3),
    "dxo_mgpu_isihara": (C.c_int, [_P, C.POINTER(IsiharaParams), C.c_int64, C.c_int] + [C.POINTER(_P)] * 3),
    "dxo_mgpu_heat": (C.c_int, [_P, C.c_double, C.c_double, C.c_int, C.c_int64, C.c_int] + [C.POINTER(_P)] * 5),
    "dxo_stream_probe": (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P, _P]),
}

_lib = None
_lib_lock = threading.Lock()


class DxoError(RuntimeError):
    pass


def _share_hip_runtime_with_torch() -> None:
    """One HIP/HSA runtime per process.

    PyTorch-ROCm wheels bundle their own `libamdhip64.so` (SONAME libamdhip64.so.7) and load it by the
    unversioned name, so if libdxo_hip.so pulled in /opt/rocm's copy first, a later `import torch` would
    load a SECOND runtime whose hsa_init sees no GPU ("No HIP GPUs are available"). When torch is
    installed but not yet imported, map its bundled runtime first: libdxo_hip.so's NEEDED
    `libamdhip64.so.7` then resolves to that same object (SONAME match) and torch later finds its own file
    already mapped. Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = pathlib.Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path: str | pathlib.Path | None = None) -> C.CDLL:
    """dlopen libdxo_hip.so and type every entry point. Raises if the library is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        import os

        override = os.environ.get("DXO_HIP_LIBRARY")   # experiments: another build of the same ABI (scripts/exp)
        p = pathlib.Path(path) if path is not None else (pathlib.Path(override) if override else LIB_PATH)
        if not p.exists():
            raise DxoError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the quadrature-point kernels."
            )
        _share_hip_runtime_with_torch()
        lib = C.CDLL(str(p))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = ABI mismatch, let it propagate
            fn.restype = res
            fn.argtypes = args
        if lib.dxo_abi_version() != ABI_VERSION:
            raise DxoError(f"ABI version mismatch: library {lib.dxo_abi_version()}, binding {ABI_VERSION}")
        if path is None:
            _lib = lib
        return lib


def declared_symbols() -> list[str]:
    return list(_SIGNATURES)


def _ptr(a) -> int | None:
    """Address of a NumPy array, a raw int address, or None."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return int(a)
    return a.ctypes.data


class _PinnedPool:
    """Size-keyed free lists of hipHostMalloc blocks. A block handed out as an ndarray comes back through a
    weakref finalizer on the ctypes object that every NumPy view of it keeps alive (ndarray.base chain), i.e.
    only when no view can observe a later overwrite. After close() returning blocks are freed instead.

    The finalizer can run at ANY allocation point of ANY thread (a cyclic-GC pass collecting a dead result array), also
    while that same thread is inside empty() — so `_give_back` takes no lock at all: it appends to a deque (atomic in
    CPython), and the deque is drained into the free lists by empty() / close() under the lock. hipHostFree, which may
    block, is never called with the lock held."""

    KEEP_PER_SIZE = 4

    def __init__(self, lib):
        self.lib = lib
        self.free: dict[int, list[int]] = {}
        self.returned: collections.deque = collections.deque()   # (addr, cap) handed back by finalizers, not yet sorted in
        self.closed = False
        self.lock = threading.RLock()

    def _drain(self, doomed: list) -> None:
        """Sort the returned blocks into the free lists (lock held); surplus ones go to `doomed` for the caller to free."""
        while True:
            try:
                addr, cap = self.returned.popleft()
            except IndexError:
                return
            lst = self.free.get(cap)
            if lst is None:
                lst = self.free[cap] = []
            if self.closed or len(lst) >= self.KEEP_PER_SIZE:
                doomed.append(addr)
            else:
                lst.append(addr)

    def empty(self, n: int, dtype: np.dtype) -> np.ndarray:
        cap = max(n * dtype.itemsize, 1)
        doomed: list = []
        with self.lock:
            if self.closed:
                raise DxoError("pinned pool used after Context.close()")
            self._drain(doomed)
            lst = self.free.get(cap)
            addr = lst.pop() if lst else None
            if addr is None:
                # the batch size changed (or first call): drop idle blocks of other sizes before growing
                for other in [k for k in self.free if k != cap]:
                    doomed.extend(self.free.pop(other))
        for a in doomed:
            self.lib.dxo_host_free(None, _P(a))
        if addr is None:
            p = _P()
            rc = self.lib.dxo_host_alloc(None, cap, C.byref(p))
            if rc != 0:
                raise DxoError(f"hipHostMalloc({cap}) failed: {ERRORS.get(rc, rc)}")
            addr = p.value
        buf = (C.c_char * cap).from_address(addr)
        fin = weakref.finalize(buf, self._give_back, addr, cap)
        fin.atexit = False   # at interpreter exit the OS reclaims the mapping
        return np.frombuffer(buf, dtype=dtype, count=n)

    def _give_back(self, addr: int, cap: int) -> None:
        # lock-free on purpose (see the class docstring). After close() nobody drains any more: free right here.
        if self.closed:
            self.lib.dxo_host_free(None, _P(addr))
            return
        self.returned.append((addr, cap))
        if self.closed:      # close() finished both of its drains between the test above and the append: nobody drains again
            doomed: list = []
            self._drain(doomed)          # popleft is atomic: a block is freed by exactly one of the racing drains
            for a in doomed:
                self.lib.dxo_host_free(None, _P(a))

    def close(self) -> None:
        doomed: list = []
        with self.lock:
            self.closed = True
            self._drain(doomed)
            for lst in self.free.values():
                doomed.extend(lst)
            self.free.clear()
        for a in doomed:
            self.lib.dxo_host_free(None, _P(a))
        doomed = []
        self._drain(doomed)   # a finalizer that raced with the flag
        for a in doomed:
            self.lib.dxo_host_free(None, _P(a))


class Context:
    """One dxo_ctx: a (process, GPU) pair owning streams and device scratch."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.dxo_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise DxoError(
                f"dxo_ctx_create(device={device}) failed with {ERRORS.get(rc, rc)}: no usable MI355X/HIP device. "
                "The HIP path is the only implementation; it does not fall back to the CPU."
            )
        self._h = h
        self.device = int(device)
        self._pinned: list[tuple[int, np.ndarray]] = []
        self._pool = _PinnedPool(self.lib)
        self._lock = threading.RLock()   # dxo_ctx itself also serialises its entry points (include/dxo.h)

    @classmethod
    def borrow(cls, handle: int, device: int = 0) -> "Context":
        """A Context over a dxo_ctx that something else owns (the contexts of a MultiGpu group): close() releases what
        this wrapper allocated (pinned buffers) but does not destroy the dxo_ctx."""
        self = cls.__new__(cls)
        self.lib = load_library()
        self._h = _P(handle)
        self.device = int(device)
        self._pinned = []
        self._pool = _PinnedPool(self.lib)
        self._lock = threading.RLock()
        self._borrowed = True
        return self

    # -- plumbing ----------------------------------------------------------------------------
    def check(self, rc: int, what: str) -> None:
        if rc == 0:
            return
        msg = self.lib.dxo_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc < 0:
            raise ValueError(f"{what}: {ERRORS.get(rc, rc)}: {msg}")
        raise DxoError(f"{what}: HIP error {rc}: {msg}")

    def close(self) -> None:
        if getattr(self, "_h", None):
            for addr, _ in self._pinned:
                self.lib.dxo_host_free(self._h, _P(addr))
            self._pinned.clear()
            self._pool.close()
            if not getattr(self, "_borrowed", False):
                self.lib.dxo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, key: str, value: int) -> None:
        self.check(self.lib.dxo_ctx_set_option(self._h, key.encode(), int(value)), f"set_option({key})")
        self.__dict__.setdefault("_opt_mirror", {})[key] = int(value)

    def option_is(self, key: str, value: int) -> bool:
        """Is the option known (from this binding's own set_option calls) to hold `value` already? Saves the per-call pair of
        set_option round trips on latency-bound batches; an option never set through the binding counts as its library default."""
        m = self.__dict__.get("_opt_mirror")
        return (m.get(key, None) if m else None) == int(value)

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        self.check(self.lib.dxo_ctx_get_option(self._h, key.encode(), C.byref(v)), f"get_option({key})")
        return v.value

    def set_stream(self, stream_handle: int | None) -> None:
        self.check(self.lib.dxo_ctx_set_stream(self._h, _P(stream_handle)), "set_stream")

    def synchronize(self) -> None:
        self.check(self.lib.dxo_ctx_synchronize(self._h), "synchronize")

    def device_info(self) -> dict:
        info = DeviceInfo()
        self.check(self.lib.dxo_ctx_device_info(self._h, C.byref(info)), "device_info")
        return {"name": info.name.decode(), "arch": info.arch.decode(), "compute_units": info.compute_units,
                "wavefront_size": info.wavefront_size, "total_mem_bytes": info.total_mem_bytes}

    def last_timing(self) -> dict:
        t = Timing()
        self.check(self.lib.dxo_last_timing(self._h, C.byref(t)), "last_timing")
        return {"h2d_ms": t.h2d_ms, "kernel_ms": t.kernel_ms, "d2h_ms": t.d2h_ms, "total_ms": t.total_ms}

    def pinned_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """NumPy array backed by hipHostMalloc memory owned by this context (freed by pinned_free or close():
        the caller manages the lifetime; for buffers that are handed to users see pinned_recycled)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
        p = _P()
        self.check(self.lib.dxo_host_alloc(self._h, n * dtype.itemsize, C.byref(p)), "host_alloc")
        buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
        self._pinned.append((p.value, arr))
        return arr

    def pinned_free(self, arr: np.ndarray) -> None:
        """Release a buffer obtained from pinned_empty (the array must not be used afterwards)."""
        addr = arr.ctypes.data
        for k, (a, _) in enumerate(self._pinned):
            if a == addr:
                self._pinned.pop(k)
                self.check(self.lib.dxo_host_free(self._h, _P(addr)), "host_free")
                return

    def pinned_recycled(self, n: int, dtype=np.float64) -> np.ndarray:
        """A flat pinned (hipHostMalloc) array whose memory returns to this context's pool only once the array
        AND every view of it have been garbage-collected — so it can be handed to a user like a fresh ndarray
        (the reference's kernels return fresh arrays, demo_plasticity_von_mises.py:352) while D2H copies still
        land in page-locked memory that has been touched before (a first-touched pageable 344 MB array costs
        ~30 ms of page faults inside the copy). No buffer is ever freed or reused while a view of it is alive."""
        return self._pool.empty(int(n), np.dtype(dtype))

    # -- kernels -------------------------------------------------------------------------------
    def von_mises(self, prm: VmParams, d: int, n: int, mem: int, deps, sigma_n, p, C_tang, sigma, dp) -> None:
        rc = self.lib.dxo_von_mises(self._h, C.byref(prm), int(d), int(n), int(mem), _ptr(deps), _ptr(sigma_n),
                                    _ptr(p), _ptr(C_tang), _ptr(sigma), _ptr(dp))
        self.check(rc, "dxo_von_mises")

    def pin(self, array: np.ndarray) -> np.ndarray:
        """Page-lock an array the caller owns (dxo_host_register) — typically the `x.array` of the coefficient a factory's
        `outputs=` writes into, once, at set-up. The array must stay alive until `unpin(array)` or the end of the process."""
        a = np.asarray(array)
        if not a.flags["C_CONTIGUOUS"]:
            raise ValueError("pin: the array must be C-contiguous")
        self.check(self.lib.dxo_host_register(self._h, _P(a.ctypes.data), int(a.nbytes)), "dxo_host_register")
        return a

    def unpin(self, array: np.ndarray) -> None:
        self.check(self.lib.dxo_host_unregister(self._h, _P(np.asarray(array).ctypes.data)), "dxo_host_unregister")

    def vm_state(self, d: int, n: int) -> "VmState":
        """Device mirror of the von Mises history variables for n points (dxo_vm_state_*, include/dxo.h)."""
        return VmState(self, d, n)

    def mohr_coulomb(self, prm: McParams, n: int, mem: int, deps, sigma_n, C_tang, sigma, niter=None, yielding=None,
                     norm_res=None, dlambda=None) -> None:
        rc = self.lib.dxo_mohr_coulomb(self._h, C.byref(prm), int(n), int(mem), _ptr(deps), _ptr(sigma_n), _ptr(C_tang),
                                       _ptr(sigma), _ptr(niter), _ptr(yielding), _ptr(norm_res), _ptr(dlambda))
        self.check(rc, "dxo_mohr_coulomb")

    def mc_state(self, n: int) -> "McState":
        """Device mirror of the Mohr-Coulomb history variable sigma_n for n points (dxo_mc_state, include/dxo.h)."""
        return McState(self, n)

    def mohr_coulomb_field(self, prm: McParams, mesh_handle, mem: int, u, sigma_n, C_tang, sigma, niter=None, yielding=None,
                           norm_res=None, dlambda=None) -> None:
        rc = self.lib.dxo_mohr_coulomb_field(self._h, C.byref(prm), mesh_handle, int(mem), _ptr(u), _ptr(sigma_n), _ptr(C_tang),
                                             _ptr(sigma), _ptr(niter), _ptr(yielding), _ptr(norm_res), _ptr(dlambda))
        self.check(rc, "dxo_mohr_coulomb_field")

    def icnn_field(self, model: int, precision: int, mesh_handle, mem: int, u, dP, P) -> None:
        self.check(self.lib.dxo_icnn_field(self._h, _P(model), int(precision), mesh_handle, int(mem), _ptr(u), _ptr(dP), _ptr(P)),
                   "dxo_icnn_field")

    def isihara_field(self, prm: "IsiharaParams", mesh_handle, mem: int, u, dP, P) -> None:
        self.check(self.lib.dxo_isihara_field(self._h, C.byref(prm), mesh_handle, int(mem), _ptr(u), _ptr(dP), _ptr(P)),
                   "dxo_isihara_field")

    def conductivity(self, A: float, B: float, n: int, mem: int, T, k, dkdT) -> None:
        self.check(self.lib.dxo_conductivity(self._h, float(A), float(B), int(n), int(mem), _ptr(T), _ptr(k), _ptr(dkdT)),
                   "dxo_conductivity")

    def mc_summary(self, n: int, niter, yielding=None, norm_res=None, nbins: int = 201) -> dict:
        """Device-side inner-Newton summary (the reference's printout, demo_plasticity_mohr_coulomb.py:584-591).
        niter / yielding / norm_res are DEVICE pointers (or objects with data_ptr())."""
        def dp(a):
            return a.data_ptr() if hasattr(a, "data_ptr") else a
        hist = np.zeros(nbins, dtype=np.int64)
        my, mr = C.c_double(), C.c_double()
        nans = np.zeros(2, dtype=np.int64)
        rc = self.lib.dxo_mc_summary(self._h, int(n), _ptr(dp(niter)), _ptr(dp(yielding)), _ptr(dp(norm_res)), int(nbins),
                                     _ptr(hist), C.byref(my), C.byref(mr), _ptr(nans))
        self.check(rc, "dxo_mc_summary")
        nz = np.flatnonzero(hist)
        return {"unique_iters": nz.astype(np.int32), "counts": hist[nz], "max_yielding": my.value, "max_norm_res": mr.value,
                "nan_yielding": int(nans[0]), "nan_norm_res": int(nans[1])}

    # state_dict key -> dxo_icnn_weights field (shapes as torch stores them)
    ICNN_KEYS = {
        "layers.0.weight": ("layers0_weight", (64, 3)), "layers.0.bias": ("layers0_bias", (64,)),
        "layers.1.weights": ("layers1_weights", (64, 64)), "skip_layers.1.weight": ("skip1_weight", (64, 3)),
        "skip_layers.1.bias": ("skip1_bias", (64,)), "layers.2.weights": ("layers2_weights", (64, 64)),
        "skip_layers.2.weight": ("skip2_weight", (64, 3)), "skip_layers.2.bias": ("skip2_bias", (64,)),
        "layers.3.weights": ("layers3_weights", (1, 64)), "skip_layers.3.weights": ("skip3_weights", (1, 3)),
    }

    def icnn_create(self, state_dict) -> int:
        """Upload an ICNN from a mapping {state_dict key: array-like} (keys with '.' or '__'). Returns a handle."""
        w = IcnnWeights()
        keep = []
        norm = {str(k).replace("__", "."): v for k, v in state_dict.items()}
        for key, (field, shape) in self.ICNN_KEYS.items():
            if key not in norm:
                raise ValueError(f"ICNN state_dict lacks {key!r}")
            v = norm[key]
            v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(v, dtype=np.float32)
            if a.shape != shape:
                raise ValueError(f"{key}: shape {a.shape}, expected {shape} (the demo's 3-64-64-64-1 network)")
            keep.append(a)
            setattr(w, field, a.ctypes.data)
        w.n_hidden = 64
        h = _P()
        self.check(self.lib.dxo_icnn_create(self._h, C.byref(w), C.byref(h)), "dxo_icnn_create")
        return h.value

    def icnn_destroy(self, model: int) -> None:
        if self._h and model:
            self.lib.dxo_icnn_destroy(self._h, _P(model))

    def icnn_correction(self, model: int) -> np.ndarray:
        out = np.empty(4)
        self.check(self.lib.dxo_icnn_correction(self._h, _P(model), _ptr(out)), "dxo_icnn_correction")
        return out

    def icnn_eval(self, model: int, precision: int, n: int, mem: int, F, dP, P) -> None:
        rc = self.lib.dxo_icnn_eval(self._h, _P(model), int(precision), int(n), int(mem), _ptr(F), _ptr(dP), _ptr(P))
        self.check(rc, "dxo_icnn_eval")

    def isihara(self, prm: "IsiharaParams", n: int, mem: int, F, dP, P) -> None:
        rc = self.lib.dxo_isihara(self._h, C.byref(prm), int(n), int(mem), _ptr(F), _ptr(dP), _ptr(P))
        self.check(rc, "dxo_isihara")

    def assign(self, desc: "AssignDesc", flat_dofs, values, coeff, coeff_size: int) -> None:
        """Device pointers (ints) or objects with .ctypes — dxo_assign works on device memory only."""
        rc = self.lib.dxo_assign(self._h, C.byref(desc), _ptr(flat_dofs), _ptr(values), _ptr(coeff), int(coeff_size))
        self.check(rc, "dxo_assign")

    def assign_plan(self, desc: "AssignDesc", flat_dofs, coeff_size: int) -> "AssignPlan":
        """dxo_assign_plan_create: the ownership pass of dxo_assign done ONCE for a dofmap; `.apply(values, coeff)` is then a
        single gather (DEVICE pointers). Use it when the same operator is assigned at every Newton iteration."""
        return AssignPlan(self, desc, flat_dofs, coeff_size)

    def stream_probe(self, read_chunks: int, write_chunks: int, n_tiles: int, src, dst) -> None:
        rc = self.lib.dxo_stream_probe(self._h, int(read_chunks), int(write_chunks), int(n_tiles), _ptr(src), _ptr(dst))
        self.check(rc, "dxo_stream_probe")

    def vm_expand_tangent(self, prm: VmParams, d: int, n: int, mem: int, sigma, dp, C_tang) -> None:
        rc = self.lib.dxo_vm_expand_tangent(self._h, C.byref(prm), int(d), int(n), int(mem), _ptr(sigma), _ptr(dp),
                                            _ptr(C_tang))
        self.check(rc, "dxo_vm_expand_tangent")

    def vm_clear_marks(self, n: int, dp) -> None:
        """-0.0 -> +0.0 in a device dp array (the producer's mark of the reference's 0/0 point, option vm_mark_indeterminate)."""
        self.check(self.lib.dxo_vm_clear_marks(self._h, int(n), _ptr(dp)), "dxo_vm_clear_marks")

    def vm_commit_state(self, d: int, n: int, p, dp, sigma_n, sigma) -> None:
        self.check(self.lib.dxo_vm_commit_state(self._h, int(d), int(n), _ptr(p), _ptr(dp), _ptr(sigma_n), _ptr(sigma)),
                   "dxo_vm_commit_state")

    # -- output arena (placement-calibrated device memory, include/dxo.h "output arena") --------------
    def output_alloc(self, nbytes: int) -> int:
        """Device pointer of a block whose virtual range was chosen for streaming-write speed (dxo_output_alloc)."""
        p = _P()
        self.check(self.lib.dxo_output_alloc(self._h, int(nbytes), C.byref(p)), "dxo_output_alloc")
        return p.value

    def output_free(self, ptr: int) -> None:
        if self._h and ptr:
            self.check(self.lib.dxo_output_free(self._h, _P(ptr)), "dxo_output_free")

    def output_info(self, ptr: int) -> dict:
        info = PlacementInfo()
        self.check(self.lib.dxo_output_info(self._h, _P(ptr), C.byref(info)), "dxo_output_info")
        kinds = ["2MB_chunks" if (info.vmm_mask >> k) & 1 else "hipMalloc" for k in range(info.candidates)]
        return {"mode": {0: "hipMalloc", 2: "candidates"}[info.mode],
                "candidates": info.candidates, "chosen": info.chosen,
                "chosen_kind": kinds[info.chosen] if 0 <= info.chosen < len(kinds) else "hipMalloc",
                "kinds": kinds, "probe": {0: "store_stream", 1: "six_stream_mix", 2: "vm_tile", 3: "caller"}.get(info.probe_kind, str(info.probe_kind)),
                "tuned_blocks_per_cu": info.tuned_blocks_per_cu,
                "probe_GBps": [round(info.probe_GBps[k], 1) for k in range(info.candidates)],
                "chosen_GBps": round(info.chosen_GBps, 1), "calibration_ms": info.calibration_ms, "rounds": info.rounds}

    def output_tensors(self, sizes, dtype=None):
        """Flat torch CUDA tensors of `sizes` elements each (fp64 unless dtype is given), carved from ONE
        dxo_output_alloc block (each on a 256-byte border). The block stays alive as long as any of the tensors
        (or the context) does; `tensors[0].dxo_block.info` holds the calibration record."""
        import torch

        dtype = dtype or torch.float64
        item = torch.empty((), dtype=dtype).element_size()
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (int(n) * item + 255) // 256 * 256
        block = _ArenaBlock(self, max(total, 256))
        typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8"}[dtype]
        out = []
        for n, off in zip(sizes, offs):
            view = _CudaArrayView(block, block.ptr + off, int(n), typestr)
            t = torch.as_tensor(view, device=torch.device("cuda", self.device)) if n else torch.empty(0, dtype=dtype, device=torch.device("cuda", self.device))
            t.dxo_block = block   # keeps the arena block alive with the tensor
            out.append(t)
        return out

    def output_tensors_probed(self, sizes, launch, bytes_per_launch: float = 0.0, shapes=(0,), dtype=None):
        """`output_tensors` with the caller's own consumer as the probe (dxo_output_alloc_probed): `launch(ptrs, shape)`
        gets the device addresses the `sizes` arrays would have inside a candidate block and must enqueue one pass of the
        kernel that will write them (e.g. `ctx.heat(..., MEM_DEVICE, ...)`) without synchronising."""
        import torch

        dtype = dtype or torch.float64
        item = torch.empty((), dtype=dtype).element_size()
        offs, total = [], 0
        for m in sizes:
            offs.append(total)
            total += (int(m) * item + 255) // 256 * 256
        errors = []

        def _cb(block, shape, _user):
            try:
                launch([block + o for o in offs], shape)
            except Exception as exc:   # noqa: BLE001 — never unwind through the C frames
                errors.append(exc)

        cb = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)(_cb)
        sh = (C.c_int32 * len(shapes))(*[int(x) for x in shapes])
        p = _P()
        self.check(self.lib.dxo_output_alloc_probed(self._h, max(total, 256), C.cast(cb, _P), None, float(bytes_per_launch), sh, len(shapes),
                                                    C.byref(p)), "dxo_output_alloc_probed")
        block = _ArenaBlock(self, max(total, 256), ptr=p.value)
        if errors:
            raise errors[0]
        typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8"}[dtype]
        out = []
        for m, off in zip(sizes, offs):
            view = _CudaArrayView(block, block.ptr + off, int(m), typestr)
            t = torch.as_tensor(view, device=torch.device("cuda", self.device)) if m else torch.empty(0, dtype=dtype, device=torch.device("cuda", self.device))
            t.dxo_block = block
            out.append(t)
        return out

    def vm_output_tensors(self, n: int, d: int):
        """(C_tang, sigma, dp) for n points as flat fp64 CUDA tensors in ONE arena block calibrated with the von Mises
        kernel itself (dxo_vm_output_alloc): candidates are timed running vm_tile in two launch shapes and the pair
        (block, shape) that makes the kernel fastest is kept; the kernel picks the shape up whenever it writes there."""
        import torch

        ptrs = [_P() for _ in range(3)]
        self.check(self.lib.dxo_vm_output_alloc(self._h, int(d), int(n), *(C.byref(q) for q in ptrs)), "dxo_vm_output_alloc")
        block = _ArenaBlock(self, 0, ptr=ptrs[0].value)
        out = []
        for q, m in zip(ptrs, (n * d * d, n * d, n)):
            view = _CudaArrayView(block, q.value, int(m), "<f8")
            t = torch.as_tensor(view, device=torch.device("cuda", self.device)) if m else torch.empty(0, dtype=torch.float64, device=torch.device("cuda", self.device))
            t.dxo_block = block
            out.append(t)
        return out

    def device_alloc(self, nbytes: int) -> int:
        p = _P()
        self.check(self.lib.dxo_device_alloc(self._h, int(nbytes), C.byref(p)), "dxo_device_alloc")
        return p.value

    def device_free(self, ptr: int) -> None:
        if self._h and ptr:
            self.lib.dxo_device_free(self._h, _P(ptr))

    def copy(self, dst, src, nbytes: int, kind: int) -> None:
        self.check(self.lib.dxo_copy(self._h, _ptr(dst), _ptr(src), int(nbytes), int(kind)), "dxo_copy")

    def heat(self, A: float, B: float, gdim: int, n: int, mem: int, T, sigma, q, dqdT, dqdsigma) -> None:
        rc = self.lib.dxo_heat(self._h, float(A), float(B), int(gdim), int(n), int(mem), _ptr(T), _ptr(sigma),
                               _ptr(q), _ptr(dqdT), _ptr(dqdsigma))
        self.check(rc, "dxo_heat")


class _ArenaBlock:
    """Owner object of one dxo_output_alloc block; freed when the last tensor view and this object are gone."""

    def __init__(self, ctx: Context, nbytes: int, ptr: int | None = None):
        self.ctx = ctx
        self.nbytes = nbytes
        self.ptr = ctx.output_alloc(nbytes) if ptr is None else ptr   # ptr: a block the library has already handed out
        self.info = ctx.output_info(self.ptr)
        self._fin = weakref.finalize(self, _ArenaBlock._release, weakref.ref(ctx), self.ptr)
        self._fin.atexit = False

    @staticmethod
    def _release(ctx_ref, ptr):
        ctx = ctx_ref()
        if ctx is not None and ctx._h:
            try:
                ctx.output_free(ptr)
            except Exception:
                pass   # the context (and with it every arena block) is already gone


class AssignPlan:
    """dxo_assign_plan: per coefficient entry the position in `values` of the entry NumPy's sequential assignment leaves there."""

    def __init__(self, ctx: "Context", desc: "AssignDesc", flat_dofs, coeff_size: int):
        self.ctx, self.coeff_size = ctx, int(coeff_size)
        h = _P()
        ctx.check(ctx.lib.dxo_assign_plan_create(ctx._h, C.byref(desc), _ptr(flat_dofs), self.coeff_size, C.byref(h)), "dxo_assign_plan_create")
        self._h = h
        self._fin = weakref.finalize(self, AssignPlan._destroy, ctx, h)

    @staticmethod
    def _destroy(ctx, h):
        ctx.lib.dxo_assign_plan_destroy(ctx._h, h)      # a closed context passes NULL: the plan's device block is still freed

    def apply(self, values, coeff) -> None:
        self.ctx.check(self.ctx.lib.dxo_assign_apply(self.ctx._h, self._h, _ptr(values), _ptr(coeff)), "dxo_assign_apply")

    def form(self) -> dict:
        """dxo_assign_plan_form: which order the plan is applied in (0 = a large plan before its first apply, 1 = by coefficient entry,
        2 = by position in `values`) and the two timings its first apply took."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        f = self.ctx.lib.dxo_assign_plan_form(self._h, C.byref(a), C.byref(b))
        return {"form": int(f), "ms_dof_order": a.value, "ms_source_order": b.value}

    def close(self) -> None:
        self._fin()


class _CudaArrayView:
    """Minimal __cuda_array_interface__ carrier so torch can wrap library-owned device memory without a copy."""

    def __init__(self, owner, ptr: int, n: int, typestr: str):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


GATHER_NONE, GATHER_FULL, GATHER_COMPACT, GATHER_COMPACT_DIRECT, GATHER_COMPACT_PIPELINED = 0, 1, 2, 3, 4
MGPU_ID_BYTES = 128


class VmState:
    """dxo_vm_state: sigma_n, p (and the last call's sigma, dp) resident on the context's GPU.

    upload(sigma_n, p) once and after any change of the caller's arrays other than the load-step update;
    call(...) = dxo_von_mises_state; commit() = `p += dp; sigma_n <- sigma` on the device
    (demo_plasticity_von_mises.py:564-565)."""

    def __init__(self, ctx: "Context", d: int, n: int):
        self.ctx, self.d, self.n = ctx, int(d), int(n)
        h = C.c_void_p()
        ctx.check(ctx.lib.dxo_vm_state_create(ctx._h, self.d, self.n, C.byref(h)), "dxo_vm_state_create")
        self._h = h
        self._fin = weakref.finalize(self, VmState._destroy, ctx, h)

    @staticmethod
    def _destroy(ctx, h):
        ctx.lib.dxo_vm_state_destroy(ctx._h, h)   # with a closed context (NULL) the library still frees the device block

    def close(self) -> None:
        self._fin()

    def upload(self, sigma_n, p, mem: int = MEM_HOST) -> None:
        self.ctx.check(self.ctx.lib.dxo_vm_state_upload(self.ctx._h, self._h, int(mem), _ptr(sigma_n), _ptr(p)), "dxo_vm_state_upload")

    def download(self, sigma_n=None, p=None, mem: int = MEM_HOST):
        if mem == MEM_HOST:
            sigma_n = np.empty(self.n * self.d) if sigma_n is None else sigma_n
            p = np.empty(self.n) if p is None else p
        self.ctx.check(self.ctx.lib.dxo_vm_state_download(self.ctx._h, self._h, int(mem), _ptr(sigma_n), _ptr(p)), "dxo_vm_state_download")
        return sigma_n, p

    def commit(self) -> None:
        self.ctx.check(self.ctx.lib.dxo_vm_state_commit(self.ctx._h, self._h), "dxo_vm_state_commit")

    def pointers(self) -> dict:
        out = [C.c_void_p() for _ in range(4)]
        self.ctx.check(self.ctx.lib.dxo_vm_state_pointers(self.ctx._h, self._h, *(C.byref(o) for o in out)), "dxo_vm_state_pointers")
        return dict(zip(("sigma_n", "p", "sigma", "dp"), (o.value for o in out)))

    def call(self, prm: VmParams, mem: int, deps, C_tang, sigma=None, dp=None) -> None:
        rc = self.ctx.lib.dxo_von_mises_state(self.ctx._h, C.byref(prm), self._h, int(mem), _ptr(deps), _ptr(C_tang), _ptr(sigma), _ptr(dp))
        self.ctx.check(rc, "dxo_von_mises_state")

    def call_field(self, prm: VmParams, mesh_handle, mem: int, u, C_tang, sigma=None, dp=None) -> None:
        rc = self.ctx.lib.dxo_von_mises_field_state(self.ctx._h, C.byref(prm), mesh_handle, self._h, int(mem), _ptr(u), _ptr(C_tang),
                                                    _ptr(sigma), _ptr(dp))
        self.ctx.check(rc, "dxo_von_mises_field_state")


class McState:
    """dxo_mc_state: device mirror of the Mohr-Coulomb history variable sigma_n plus the stress of the last call."""

    def __init__(self, ctx: "Context", n: int):
        self.ctx, self.n = ctx, int(n)
        h = _P()
        ctx.check(ctx.lib.dxo_mc_state_create(ctx._h, self.n, C.byref(h)), "dxo_mc_state_create")
        self._h = h
        self._fin = weakref.finalize(self, McState._destroy, ctx, h)

    @staticmethod
    def _destroy(ctx, h):
        ctx.lib.dxo_mc_state_destroy(ctx._h, h)   # with a closed context (NULL) the library still frees the device block

    def close(self) -> None:
        self._fin()

    def upload(self, sigma_n, mem: int = MEM_HOST) -> None:
        self.ctx.check(self.ctx.lib.dxo_mc_state_upload(self.ctx._h, self._h, int(mem), _ptr(sigma_n)), "dxo_mc_state_upload")

    def download(self, sigma_n=None, mem: int = MEM_HOST):
        if mem == MEM_HOST and sigma_n is None:
            sigma_n = np.empty(self.n * 4)
        self.ctx.check(self.ctx.lib.dxo_mc_state_download(self.ctx._h, self._h, int(mem), _ptr(sigma_n)), "dxo_mc_state_download")
        return sigma_n

    def commit(self) -> None:
        self.ctx.check(self.ctx.lib.dxo_mc_state_commit(self.ctx._h, self._h), "dxo_mc_state_commit")

    def pointers(self) -> dict:
        out = [C.c_void_p() for _ in range(2)]
        self.ctx.check(self.ctx.lib.dxo_mc_state_pointers(self.ctx._h, self._h, *(C.byref(o) for o in out)), "dxo_mc_state_pointers")
        return dict(zip(("sigma_n", "sigma"), (o.value for o in out)))

    def call(self, prm: McParams, mem: int, deps, C_tang, sigma=None, niter=None, yielding=None, norm_res=None, dlambda=None) -> None:
        rc = self.ctx.lib.dxo_mohr_coulomb_state(self.ctx._h, C.byref(prm), self._h, int(mem), _ptr(deps), _ptr(C_tang), _ptr(sigma),
                                                 _ptr(niter), _ptr(yielding), _ptr(norm_res), _ptr(dlambda))
        self.ctx.check(rc, "dxo_mohr_coulomb_state")


class MultiGpu:
    """dxo_mgpu: cell-block sharding with the RCCL all-gather inside the library (include/dxo.h "multi-GPU").

    MultiGpu(devices=[0, 1, ...])                       one process driving several GPUs (ncclCommInitAll)
    MultiGpu.from_rank(ctx, unique_id, rank, world)     one process per GPU; `unique_id = MultiGpu.unique_id()` on
                                                        rank 0, broadcast by the caller (128 bytes)
    MultiGpu.local(devices=[0, 1, ...])                 contexts only, no communicator, RCCL never loaded: for
                                                        `von_mises_host` (NumPy arrays sharded over the GPUs' PCIe links)
    Pointer-list arguments take one device pointer (int or tensor with data_ptr()) per LOCAL device.

    Buffers handed to a collective (all_gather, von_mises with a gather) must be ordinary hipMalloc memory (torch tensors
    are): an output-arena block built from 2 MB physical chunks is accessible from its own device only and cannot be
    exported to a peer. The group's contexts therefore have "placement_vmm" = 0 (from_rank sets it on the context it is
    given), and the collectives raise ValueError (DXO_E_MEM) for a pointer inside a chunk-backed block."""

    def __init__(self, devices=None, n_dev: int | None = None, _handle=None, _ctx=None):
        self.lib = load_library()
        self._keep_ctx = _ctx
        self._borrowed: dict[int, Context] = {}
        if _handle is not None:
            self._h = _handle
            return
        if devices is None:
            devices = list(range(int(n_dev or 1)))
        arr = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = _P()
        rc = self.lib.dxo_mgpu_create(arr, len(devices), C.byref(h))
        if rc != 0:
            raise DxoError(f"dxo_mgpu_create({list(devices)}) failed with {ERRORS.get(rc, rc)}: needs that many MI355X and "
                           "RCCL (librccl.so.1); there is no CPU fallback")
        self._h = h

    @classmethod
    def local(cls, devices) -> "MultiGpu":
        lib = load_library()
        arr = (C.c_int * len(devices))(*[int(x) for x in devices])
        h = _P()
        rc = lib.dxo_mgpu_create_local(arr, len(devices), C.byref(h))
        if rc != 0:
            raise DxoError(f"dxo_mgpu_create_local({list(devices)}) failed with {ERRORS.get(rc, rc)}: no such MI355X; there is no CPU fallback")
        return cls(_handle=h)

    def set_option(self, key: str, value: int) -> None:
        """The option on every local context."""
        for i in range(self.local_count):
            self._check(self.lib.dxo_ctx_set_option(_P(self.ctx_handle(i)), key.encode(), int(value)), f"set_option({key})")

    def von_mises_host(self, prm: VmParams, d: int, n: int, deps, sigma_n, p, C_tang, sigma, dp) -> None:
        """dxo_mgpu_von_mises_host: HOST arrays of all n points, one contiguous block per local device, no collective."""
        rc = self.lib.dxo_mgpu_von_mises_host(self._h, C.byref(prm), int(d), int(n), *(_ptr(a) for a in (deps, sigma_n, p, C_tang, sigma, dp)))
        self._check(rc, "dxo_mgpu_von_mises_host")

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(MGPU_ID_BYTES)
        rc = load_library().dxo_mgpu_unique_id(buf)
        if rc != 0:
            raise DxoError(f"dxo_mgpu_unique_id failed with {ERRORS.get(rc, rc)}")
        return buf.raw

    @classmethod
    def from_rank(cls, ctx: Context, unique_id: bytes, rank: int, world: int) -> "MultiGpu":
        if len(unique_id) != MGPU_ID_BYTES:
            raise ValueError(f"unique id must be {MGPU_ID_BYTES} bytes")
        h = _P()
        buf = C.create_string_buffer(bytes(unique_id), MGPU_ID_BYTES)
        ctx.check(ctx.lib.dxo_mgpu_create_rank(ctx._h, buf, int(rank), int(world), C.byref(h)), "dxo_mgpu_create_rank")
        return cls(_handle=h, _ctx=ctx)

    def _check(self, rc: int, what: str) -> None:
        if rc == 0:
            return
        msg = self.lib.dxo_mgpu_last_error(self._h)
        msg = msg.decode() if msg else ""
        if rc < 0:
            raise ValueError(f"{what}: {ERRORS.get(rc, rc)}: {msg}")
        raise DxoError(f"{what}: error {rc}: {msg}")

    @property
    def world(self) -> int:
        return self.lib.dxo_mgpu_size(self._h)

    @property
    def local_count(self) -> int:
        return self.lib.dxo_mgpu_local_count(self._h)

    def rank(self, i: int = 0) -> int:
        return self.lib.dxo_mgpu_rank(self._h, int(i))

    def ctx_handle(self, i: int = 0) -> int:
        return self.lib.dxo_mgpu_ctx(self._h, int(i))

    def context(self, i: int = 0) -> Context:
        """The context of local device i as a (borrowed) Context: options, output arena, single-GPU entry points. Arena
        blocks of a group's contexts are hipMalloc memory ("placement_vmm" = 0): they may become RCCL buffers. The wrapper
        knows the device its dxo_ctx lives on (option "device") and dies with the group: close() releases what it allocated
        and clears its handle, so arena blocks / states made through it must be dropped before the group is closed."""
        i = int(i)
        c = self._borrowed.get(i)
        if c is None or c._h is None:
            c = Context.borrow(self.ctx_handle(i), device=self.ctx_option(i, "device"))
            self._borrowed[i] = c
        return c

    def ctx_option(self, i: int, key: str) -> int:
        v = C.c_int64()
        self._check(self.lib.dxo_ctx_get_option(_P(self.ctx_handle(i)), key.encode(), C.byref(v)), f"get_option({key})")
        return v.value

    def set_stream(self, i: int, stream_handle) -> None:
        """Launch stream of local device i (e.g. torch.cuda.current_stream(dev).cuda_stream)."""
        rc = self.lib.dxo_ctx_set_stream(_P(self.ctx_handle(i)), _P(stream_handle))
        self._check(rc, "dxo_ctx_set_stream")

    @staticmethod
    def _ptrs(seq):
        vals = [(x.data_ptr() if hasattr(x, "data_ptr") else x) for x in seq]
        return (_P * len(vals))(*[_P(v) for v in vals])

    def von_mises(self, prm: VmParams, d: int, n_per_rank: int, gather: int, deps, sigma_n, p, C_tang, sigma, dp) -> None:
        args = [self._ptrs(a) for a in (deps, sigma_n, p, C_tang, sigma, dp)]
        for a in args:
            if len(a) != self.local_count:
                raise ValueError(f"every pointer list needs {self.local_count} entries (one per local device)")
        self._check(self.lib.dxo_mgpu_von_mises(self._h, C.byref(prm), int(d), int(n_per_rank), int(gather), *args), "dxo_mgpu_von_mises")

    def _opt_ptrs(self, seq):
        """A pointer list, or NULL for an output that is not requested (None)."""
        if seq is None:
            return None
        a = self._ptrs(seq)
        if len(a) != self.local_count:
            raise ValueError(f"every pointer list needs {self.local_count} entries (one per local device)")
        return a

    def mohr_coulomb(self, prm: McParams, n_per_rank: int, gather: int, deps, sigma_n, C_tang, sigma, niter=None, yielding=None,
                     norm_res=None, dlambda=None) -> None:
        """dxo_mgpu_mohr_coulomb: every local device's cell block + (gather = GATHER_FULL) one all-gather per output."""
        args = [self._opt_ptrs(a) for a in (deps, sigma_n, C_tang, sigma, niter, yielding, norm_res, dlambda)]
        self._check(self.lib.dxo_mgpu_mohr_coulomb(self._h, C.byref(prm), int(n_per_rank), int(gather), *args), "dxo_mgpu_mohr_coulomb")

    def icnn(self, models, precision: int, n_per_rank: int, gather: int, F, dP, P) -> None:
        """dxo_mgpu_icnn; models: one dxo_icnn handle per local device (created on that device's context)."""
        m = (_P * len(models))(*[_P(x) for x in models])
        args = [self._opt_ptrs(a) for a in (F, dP, P)]
        self._check(self.lib.dxo_mgpu_icnn(self._h, m, int(precision), int(n_per_rank), int(gather), *args), "dxo_mgpu_icnn")

    def isihara(self, prm: "IsiharaParams", n_per_rank: int, gather: int, F, dP, P) -> None:
        args = [self._opt_ptrs(a) for a in (F, dP, P)]
        self._check(self.lib.dxo_mgpu_isihara(self._h, C.byref(prm), int(n_per_rank), int(gather), *args), "dxo_mgpu_isihara")

    def heat(self, A: float, B: float, gdim: int, n_per_rank: int, gather: int, T, sigma, q=None, dqdT=None, dqdsigma=None) -> None:
        args = [self._opt_ptrs(a) for a in (T, sigma, q, dqdT, dqdsigma)]
        self._check(self.lib.dxo_mgpu_heat(self._h, float(A), float(B), int(gdim), int(n_per_rank), int(gather), *args), "dxo_mgpu_heat")

    def all_gather(self, bufs, count_per_rank: int) -> None:
        a = self._ptrs(bufs)
        self._check(self.lib.dxo_mgpu_all_gather(self._h, a, int(count_per_rank)), "dxo_mgpu_all_gather")

    def synchronize(self) -> None:
        self._check(self.lib.dxo_mgpu_synchronize(self._h), "dxo_mgpu_synchronize")

    def close(self) -> None:
        if getattr(self, "_h", None):
            for c in getattr(self, "_borrowed", {}).values():     # before the dxo_ctx objects go: no wrapper keeps a dangling handle
                c.close()
            self._borrowed = {}
            self.lib.dxo_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx: dict[int, Context] = {}



def default_context(device: int = 0) -> Context:
    ctx = _default_ctx.get(device)
    if ctx is None or ctx._h is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx
