#!/usr/bin/env python3
"""Build a variant of libdxo_hip.so with extra -D flags on ONE translation unit (CPU container; hipcc cross-compiles):
    python scripts/exp/build_variant.py <name> <file.hip> [-DFOO=1 ...]
-> dolfinx_external_operator_amd/build_exp/libdxo_<name>.so (travels to the GPU box; git-ignored). The other objects are the
in-tree build's (dolfinx_external_operator_amd/build/*.o), so build the library first."""
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from dolfinx_external_operator_amd import _build as B  # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_library()
out = B.PKG / "build_exp"
out.mkdir(exist_ok=True)
obj = out / f"{name}_{pathlib.Path(src).stem}.o"
cmd = [B._hipcc(), *B.hip_flags(), *flags, "-c", str(B.CSRC / src), "-o", str(obj)]
subprocess.run(cmd, check=True)
objs = [obj if (B.PKG / "build" / (pathlib.Path(s).stem + ".o")).name == pathlib.Path(src).stem + ".o" else B.PKG / "build" / (pathlib.Path(s).stem + ".o")
        for s in B.HIP_SOURCES]
lib = out / f"libdxo_{name}.so"
subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs), "-ldl", "-lpthread"], check=True)
print(lib)
