"""Build libdxo_hip.so (HIP, gfx950) in-tree and, for tests only, the CPU oracle library.

`hipcc` cross-compiles gfx950 code objects without a GPU, so this runs in the CPU-only build
container; the resulting .so travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import hashlib
import os
import pathlib
import shutil
import subprocess

PKG = pathlib.Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
INCLUDE = ROOT / "include"
LIB = PKG / "libdxo_hip.so"
ARCH = "gfx950"

HIP_SOURCES = ["dxo_ctx.hip", "von_mises.hip", "heat.hip", "probe.hip", "mohr_coulomb.hip", "icnn.hip", "operand.hip", "vm_field.hip", "assign.hip", "heat_field.hip", "adjoint.hip", "arena.hip", "mgpu.hip", "operand_facet.hip", "field_ops.hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and pathlib.Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libdxo_hip.so cannot be built (no CPU fallback exists by design)")


def _stamp(paths, flags) -> str:
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for p in paths:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def hip_flags() -> list[str]:
    return [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
            f"-I{INCLUDE}", f"-I{CSRC}"]


# per-file flags on top of hip_flags(). icnn.hip: the SLP vectoriser must not re-pack the neuron-pair fp32 arithmetic into
# v_pk_*_f32 (a packed fp32 instruction beside a running MFMA stalls ~20 cycles, see the comment at icnn_f2)
EXTRA_FLAGS = {"icnn.hip": ["-fno-slp-vectorize"]}


def build_library(force: bool = False, verbose: bool = False) -> pathlib.Path:
    """Compile every HIP translation unit for gfx950 and link libdxo_hip.so next to this file."""
    srcs = [CSRC / s for s in HIP_SOURCES]
    deps = srcs + sorted(CSRC.glob("*.h")) + [INCLUDE / "dxo.h"]
    flags = hip_flags()
    stamp_file = PKG / ".libdxo_hip.stamp"
    stamp = _stamp(deps, flags + [f"{k}:{' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items())])
    if not force and LIB.exists() and stamp_file.exists() and stamp_file.read_text() == stamp:
        return LIB
    hipcc = _hipcc()
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    objs = []
    procs = []
    for src in srcs:
        obj = objdir / (src.stem + ".o")
        cmd = [hipcc, *flags, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{out}")
        if verbose and out.strip():
            print(out)
    tmp = LIB.with_suffix(".so.tmp")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs), "-ldl", "-lpthread"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed: {' '.join(cmd)}\n{res.stdout}")
    os.replace(tmp, LIB)
    stamp_file.write_text(stamp)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
