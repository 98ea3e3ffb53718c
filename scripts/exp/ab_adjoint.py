#!/usr/bin/env python3
"""A/B timing of builds of the consumer-side kernels (dxo_tangent_apply, dxo_tangent_diagonal, dxo_operand_adjoint) on ONE mesh
in ONE process (GPU box). usage: python scripts/exp/ab_adjoint.py [hex|tri|tet] [lib.so ...]
(default libs: the in-tree library + every build_exp/libdxo_*.so). Rounds are interleaved; results are compared with the first
library's (1e-12 of the scale)."""
import glob
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dolfinx_external_operator_amd._lib as L  # noqa: E402
from dolfinx_external_operator_amd import Context, DeviceMesh  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

cell = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] in ("hex", "tri", "tet") else "hex"
libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [str(L.LIB_PATH)] + sorted(glob.glob(str(ROOT / "dolfinx_external_operator_amd" / "build_exp" / "libdxo_*.so")))
m = (structured_mesh("hexahedron", (108,) * 3, 2, distort=0.2, seed=0) if cell == "hex" else
     structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0) if cell == "tri" else
     structured_mesh("tetrahedron", (75,) * 3, 2, distort=0.2, seed=0))
dev = torch.device("cuda:0")
bs = m.gdim
d = 4 if bs == 2 else 6
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev)
g.manual_seed(1)
A = torch.randn(npts, d, d, generator=g, device=dev, dtype=torch.float64)
Ct = (A @ A.transpose(1, 2) + torch.eye(d, device=dev, dtype=torch.float64)).reshape(-1).contiguous()   # SPD tangents
del A
S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
v = torch.randn(nn * bs, generator=g, device=dev, dtype=torch.float64)
dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)     # half the points plastic
from dolfinx_external_operator_amd import VmParams  # noqa: E402
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
stream = torch.cuda.current_stream()
runs, ref = [], None
for path in libs:
    L._lib = L.load_library(path)
    ctx = Context(0)
    ctx.set_stream(stream.cuda_stream)
    if __import__("os").environ.get("AB_ATOMICS") == "1":      # fp64 atomics into the dof vector instead of element vectors + node sums
        ctx.set_option("adjoint_atomics", 1)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    outs = {k: torch.zeros(nn * bs, dtype=torch.float64, device=dev) for k in ("apply", "diag", "force", "apply_vm", "diag_vm")}
    fns = {"apply": lambda dm=dm, o=outs["apply"]: (o.zero_(), dm.tangent_apply(Ct.data_ptr(), v.data_ptr(), o.data_ptr())),
           "diag": lambda dm=dm, o=outs["diag"]: (o.zero_(), dm.tangent_diagonal(Ct.data_ptr(), o.data_ptr())),
           "force": lambda dm=dm, o=outs["force"]: (o.zero_(), dm.adjoint("eps", bs, S.data_ptr(), o.data_ptr())),
           "apply_vm": lambda dm=dm, o=outs["apply_vm"]: (o.zero_(), dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), o.data_ptr())),
           "diag_vm": lambda dm=dm, o=outs["diag_vm"]: (o.zero_(), dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), o.data_ptr()))}
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    if ref is None:
        ref = {k: o.clone() for k, o in outs.items()}
    else:
        for k, o in outs.items():
            err = float((o - ref[k]).abs().max() / ref[k].abs().max())
            if err >= 1e-12: print(f"{path}: {k} differs from the first library by {err:.2e}")
    runs.append((pathlib.Path(path).name, fns, ctx, dm, outs, {k: [] for k in fns}))
for rnd in range(4):
    for name, fns, *_r, times in runs:
        for k, f in fns.items():
            for _ in range(2):
                f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(6):
                f()
            b.record(stream)
            torch.cuda.synchronize()
            times[k].append(a.elapsed_time(b) / 6)
for name, *_r, times in runs:
    print(json.dumps({"lib": name, "cell": cell, "points": npts, **{k + "_ms": round(statistics.median(t), 4) for k, t in times.items()}}), flush=True)
