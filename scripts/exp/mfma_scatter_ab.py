#!/usr/bin/env python3
"""Consumer-side calls on Q2 hexahedra with the element-vector contraction on the f64 matrix pipe (option adjoint_mfma = 1) against the DPP
reduce-scatter form (0). ms per call (consumer_overwrite = 1), max relative difference, bitwise reproducibility of the MFMA form."""
import json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 108
m = structured_mesh("hexahedron", (n,) * 3, 2, distort=0.2, seed=0)
dev = torch.device("cuda:0")
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
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
CT = torch.randn(npts, 36, generator=g, device=dev, dtype=torch.float64)
calls = {"apply": lambda: dm.tangent_apply(CT.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag": lambda: dm.tangent_diagonal(CT.data_ptr(), out.data_ptr()),
         "force": lambda: dm.adjoint("eps", 3, S.data_ptr(), out.data_ptr()),
         "apply_vm": lambda: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag_vm": lambda: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), out.data_ptr())}
for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["force"]):
    f = calls[name]
    rec, outs = {}, {}
    for mode in (0, 1):
        ctx.set_option("adjoint_mfma", mode)
        out.fill_(7.0); f(); torch.cuda.synchronize()
        outs[mode] = out.clone()
        if mode == 1:
            out.fill_(-3.0); f(); torch.cuda.synchronize()
            rec["mfma_bitwise_reproducible"] = bool(torch.equal(out, outs[1]))
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
        rec["mfma_ms" if mode else "dpp_ms"] = round(statistics.median(ts), 4)
    rec["max_rel_diff"] = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max())
    print(json.dumps({"cells": m.num_cells, "call": name, **rec}), flush=True)
ctx.set_option("adjoint_mfma", 0)
