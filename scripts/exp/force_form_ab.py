import json, statistics, sys
sys.path.insert(0, ".")
import torch
from dolfinx_external_operator_amd import Context, DeviceMesh
from tools.synthetic import structured_mesh
cell = sys.argv[1]
m = (structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0) if cell == "tri" else structured_mesh("tetrahedron", (75,) * 3, 2, distort=0.2, seed=0))
dev = torch.device("cuda:0"); G = m.gdim; d = 4 if G == 2 else 6
npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
g = torch.Generator(device=dev); g.manual_seed(1)
S = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64)
stream = torch.cuda.current_stream(); ctx = Context(0); ctx.set_stream(stream.cuda_stream); ctx.set_option("consumer_overwrite", 1)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
out = torch.zeros(nn * G, dtype=torch.float64, device=dev)
f = lambda: dm.adjoint("eps", G, S.data_ptr(), out.data_ptr())
rec = {}
for rnd in range(2):
    for mode in (1, 0):
        ctx.set_option("adjoint_cell", mode)
        f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            for _ in range(2): f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(8): f()
            b.record(stream); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 8)
        rec[("lane_cell" if mode else "wave_group") + str(rnd)] = round(statistics.median(ts), 4)
print(json.dumps({"cell": cell, **rec}))
