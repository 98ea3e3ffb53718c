#!/usr/bin/env python3
"""Build a variant of libdxo_hip.so with extra -D flags on ONE translation unit — or several, comma-separated — (CPU container;
hipcc cross-compiles):
    python scripts/exp/build_variant.py <name> <file.hip[,file2.hip...]> [-DFOO=1 ...]
-> dolfinx_external_operator_amd/build_exp/libdxo_<name>.so (travels to the GPU box; git-ignored). The other objects are the
in-tree build's (dolfinx_external_operator_amd/build/*.o), so build the library first.

The kernels that were measured and not shipped (scripts/exp/icnn_variants.h, adjoint_patch*.h, adjoint_variants.h) come back with
    python scripts/exp/build_variant.py experiments icnn.hip,adjoint.hip -DDXO_EXPERIMENTS
    DXO_HIP_LIBRARY=dolfinx_external_operator_amd/build_exp/libdxo_experiments.so python -m pytest tests/test_icnn.py tests/test_adjoint_gpu.py -m gpu
(their bit-identity tests run whenever the library under test accepts the options, and check the refusal otherwise)."""
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
built = {}
for one in src.split(","):
    obj = out / f"{name}_{pathlib.Path(one).stem}.o"
    extra = ["-fno-slp-vectorize"] if pathlib.Path(one).stem == "icnn" else []      # as the product build compiles that unit (_build.py)
    subprocess.run([B._hipcc(), *B.hip_flags(), *extra, *flags, "-c", str(B.CSRC / one), "-o", str(obj)], check=True)
    built[pathlib.Path(one).stem] = obj
objs = [built.get(pathlib.Path(s).stem, B.PKG / "build" / (pathlib.Path(s).stem + ".o")) for s in B.HIP_SOURCES]
lib = out / f"libdxo_{name}.so"
subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs), "-ldl", "-lpthread"], check=True)
print(lib)
