#!/usr/bin/env python3
"""Consumer-side calls with the patch form (ctx option adjoint_patch = 1: element-vector entries meet in LDS, adjoint_patch.h) against
the two-pass form (0: element vectors through HBM + node_sum), one process, one mesh. usage: python scripts/exp/patch_ab.py [hex|tri|tet]
Prints ms per call (consumer_overwrite = 1: no memset), max relative difference between the forms, and whether two runs of the
patch form give the same bits."""
import json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

cell = sys.argv[1] if len(sys.argv) > 1 else "hex"
m = (structured_mesh("hexahedron", (108,) * 3, 2, distort=0.2, seed=0) if cell == "hex" else
     structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0) if cell == "tri" else
     structured_mesh("tetrahedron", (75,) * 3, 2, distort=0.2, seed=0))
dev = torch.device("cuda:0")
bs = m.gdim
d = 4 if bs == 2 else 6
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
v = torch.randn(nn * bs, generator=g, device=dev, dtype=torch.float64)
dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
A = torch.randn(npts, d, d, generator=g, device=dev, dtype=torch.float64)
Ct = (A @ A.transpose(1, 2) + torch.eye(d, device=dev, dtype=torch.float64)).reshape(-1).contiguous()
del A
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
stream = torch.cuda.current_stream()
ctx = Context(0)
ctx.set_stream(stream.cuda_stream)
ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
out = torch.zeros(nn * bs, dtype=torch.float64, device=dev)
print(json.dumps({"patch_info": dm.patch_info()}), flush=True)
calls = {"force": lambda: dm.adjoint("eps", bs, S.data_ptr(), out.data_ptr()),
         "apply_vm": lambda: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag_vm": lambda: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), out.data_ptr()),
         "apply": lambda: dm.tangent_apply(Ct.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag": lambda: dm.tangent_diagonal(Ct.data_ptr(), out.data_ptr())}
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(calls)
res = {}
for name in only:
    f = calls[name]
    rec = {}
    outs = {}
    for mode in (0, 1):
        ctx.set_option("adjoint_patch", mode)
        out.fill_(7.0)      # overwrite mode: stale values must disappear
        f(); torch.cuda.synchronize()
        outs[mode] = out.clone()
        if mode == 1:
            out.fill_(-3.0); f(); torch.cuda.synchronize()
            rec["patch_form_bitwise_reproducible"] = bool(torch.equal(out, outs[1]))
        ts = []
        for _ in range(5):
            for _ in range(2):
                f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(8):
                f()
            b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 8)
        rec["patch_ms" if mode else "two_pass_ms"] = round(statistics.median(ts), 4)
    rec["max_rel_diff"] = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
    res[name] = rec
    print(json.dumps({"cell": cell, "points": npts, "call": name, **rec}), flush=True)
ctx.set_option("adjoint_patch", 1)
