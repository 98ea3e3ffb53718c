#!/usr/bin/env python3
"""Consumer-side calls: two-pass form (default, bit-reproducible) against fp64 hardware atomics into the dof vector (option adjoint_atomics = 1,
one pass, reproducible to rounding only). usage: python scripts/exp/atomics_ab.py [hex|tri]"""
import json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

cell = sys.argv[1] if len(sys.argv) > 1 else "hex"
m = structured_mesh("hexahedron", (108,) * 3, 2, distort=0.2, seed=0) if cell == "hex" else structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0)
dev = torch.device("cuda:0")
bs = m.gdim
d = 4 if bs == 2 else 6
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
v = torch.randn(nn * bs, generator=g, device=dev, dtype=torch.float64)
dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
stream = torch.cuda.current_stream()
ctx = Context(0)
ctx.set_stream(stream.cuda_stream)
ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
out = torch.zeros(nn * bs, dtype=torch.float64, device=dev)
calls = {"force": lambda: dm.adjoint("eps", bs, S.data_ptr(), out.data_ptr()),
         "apply_vm": lambda: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag_vm": lambda: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), out.data_ptr())}
for name, f in calls.items():
    rec, outs = {}, {}
    for mode in (0, 1):
        ctx.set_option("adjoint_atomics", mode)
        f(); torch.cuda.synchronize()
        outs[mode] = out.clone()
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
        rec["atomics_ms" if mode else "two_pass_ms"] = round(statistics.median(ts), 4)
    rec["max_rel_diff"] = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
    print(json.dumps({"cell": cell, "call": name, **rec}), flush=True)
ctx.set_option("adjoint_atomics", 0)
