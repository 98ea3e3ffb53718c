#!/usr/bin/env python3
"""The state-based consumer-side calls with the scatter on the fp64 matrix pipe (option adjoint_mfma = 1: csrc/scatter_mfma.h on P2 triangles / tetrahedra,
cell8_mfma.h on Q1 hexahedra) against the generic kernels (0), 10^7 points, two interleaved rounds.
usage: python scripts/exp/gm_scatter_ab.py tri|tet|hex1 apply_vm,diag_vm"""
import json, statistics, sys
sys.path.insert(0, ".")
import torch
from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams
from tools.synthetic import structured_mesh
cell = sys.argv[1]
m = (structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0) if cell == "tri" else structured_mesh("tetrahedron", (75,) * 3, 2, distort=0.2, seed=0) if cell == "tet"
     else structured_mesh("hexahedron", (108,) * 3, 1, distort=0.2, seed=0))      # hex1: Q1 hexahedra
dev = torch.device("cuda:0"); G = m.gdim; d = 4 if G == 2 else 6
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
v = torch.randn(nn * G, generator=g, device=dev, dtype=torch.float64)
dpv = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).clamp_(min=0.0)
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
stream = torch.cuda.current_stream(); ctx = Context(0); ctx.set_stream(stream.cuda_stream); ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
out = torch.zeros(nn * G, dtype=torch.float64, device=dev)
calls = {"apply_vm": lambda: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), out.data_ptr()),
         "diag_vm": lambda: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), out.data_ptr())}
for name in sys.argv[2].split(","):
    f = calls[name]
    rec, outs = {}, {}
    for rnd in range(2):
        for mode in (0, 1):
            ctx.set_option("adjoint_mfma", mode)
            out.fill_(7.0); f(); torch.cuda.synchronize(); outs[mode] = out.clone()
            ts = []
            for _ in range(5):
                for _ in range(2): f()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(8): f()
                b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 8)
            rec[("mfma" if mode else "lds") + str(rnd)] = round(statistics.median(ts), 4)
    out.fill_(-1.0); f(); torch.cuda.synchronize()
    rec["rel_diff"] = float((outs[1] - outs[0]).abs().max() / outs[0].abs().max()); rec["bitwise_again"] = bool(torch.equal(out, outs[1]))
    print(json.dumps({"cell": cell, "call": name, **rec}))
