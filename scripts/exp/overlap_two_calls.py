#!/usr/bin/env python3
"""Do a memory-bound node_sum and an arithmetic-bound element kernel share the chip? Two consumer-side calls (two contexts, two meshes, two
streams) issued concurrently against the same two calls back to back on one stream. If the pair costs clearly less than twice one call, a
call split into slabs whose node sums run beside the next slab's element kernel would pay. usage: overlap_two_calls.py [hex|tri]"""
import json, pathlib, statistics, sys, time
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
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ctxs, dms, outs = [], [], []
for st in streams:
    c = Context(0); c.set_stream(st.cuda_stream); c.set_option("consumer_overwrite", 1)
    ctxs.append(c); dms.append(DeviceMesh.from_synthetic(m, ctx=c)); outs.append(torch.zeros(nn * bs, dtype=torch.float64, device=dev))
calls = {"force": lambda k: dms[k].adjoint("eps", bs, S.data_ptr(), outs[k].data_ptr()),
         "apply_vm": lambda k: dms[k].tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), outs[k].data_ptr())}
for name, f in calls.items():
    f(0); f(1); torch.cuda.synchronize()
    res = {}
    for mode in ("one_stream", "two_streams"):
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                if mode == "one_stream":
                    f(0); f(0)
                else:
                    f(0); f(1)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
        res[mode + "_ms_per_pair"] = round(statistics.median(ts), 4)
    print(json.dumps({"cell": cell, "call": name, **res, "ratio": round(res["two_streams_ms_per_pair"] / res["one_stream_ms_per_pair"], 3)}), flush=True)
