#!/usr/bin/env python3
"""Consumer-side calls on Q2 hexahedra 108^3 with the brick form (ctx option adjoint_brick = 1, csrc/cell8_brick.h) against the per-cell
element-vector form (0): ms per call (consumer_overwrite = 1), max relative difference, bit-reproducibility, and the brick statistics.
usage: python scripts/exp/brick_ab.py [n]"""
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 108
m = structured_mesh("hexahedron", (n,) * 3, 2, distort=0.2, seed=0)
dev = torch.device("cuda:0")
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev)
g.manual_seed(1)
S = torch.randn(npts * 6, generator=g, device=dev, dtype=torch.float64)
v = torch.randn(nn * 3, generator=g, device=dev, dtype=torch.float64)
dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
stream = torch.cuda.current_stream()
ctx = Context(0)
ctx.set_stream(stream.cuda_stream)
ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
out = torch.zeros(nn * 3, dtype=torch.float64, device=dev)
calls = {"force": lambda: dm.adjoint("eps", 3, S.data_ptr(), out.data_ptr()),
         "apply_vm": lambda: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag_vm": lambda: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), out.data_ptr())}


def time_call(f, reps=20):
    for _ in range(3):
        f()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            f()
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return statistics.median(ts)


res = {}
for name, f in calls.items():
    row = {}
    ref = None
    for mode in (0, 1, 0, 1):
        ctx.set_option("adjoint_brick", mode)
        f()
        torch.cuda.synchronize()
        cur = out.clone()
        if mode == 0:
            ref = cur
        else:
            row["max_rel_diff"] = float((cur - ref).abs().max() / ref.abs().max())
            f()
            torch.cuda.synchronize()
            row["bit_reproducible"] = bool(torch.equal(out, cur))
        row.setdefault(f"ms_brick{mode}", []).append(round(time_call(f), 4))
    res[name] = row
    print(json.dumps({name: row}), flush=True)
ctx.close()
