#!/usr/bin/env python3
"""dxo_von_mises_residual on Q2 hexahedra (108^3 cells): the fused kernel against the two calls (option vm_residual_fused), one process,
interleaved rounds. usage: python scripts/exp/archive/residual_ab.py [lib.so ...]"""
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import dolfinx_external_operator_amd._lib as L  # noqa: E402
from dolfinx_external_operator_amd import Context, DeviceMesh, VmParams  # noqa: E402
from tools.bench_device_loop import field_dofs  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [str(L.LIB_PATH)]
m = structured_mesh("hexahedron", (108,) * 3, 2, distort=0.2, seed=0)
dev = torch.device("cuda:0")
npts, nn, d = m.num_cells * m.nq, m.node_x.shape[0], 6
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
rng = np.random.Generator(np.random.PCG64(0))
u = torch.from_numpy(field_dofs(m, rng)).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
sn = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64) * 100
pp = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
sig, dp = torch.zeros(npts * d, dtype=torch.float64, device=dev), torch.zeros(npts, dtype=torch.float64, device=dev)
R = torch.zeros(nn * 3, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream()
runs = []
for path in libs:
    L._lib = L.load_library(path)
    ctx = Context(0)
    ctx.set_stream(stream.cuda_stream)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    for fused in (1, 0):
        def fn(ctx=ctx, dm=dm, fused=fused):
            ctx.set_option("vm_residual_fused", fused)
            R.zero_()
            dm.von_mises_residual(prm, u.data_ptr(), sn.data_ptr(), pp.data_ptr(), sig.data_ptr(), dp.data_ptr(), R.data_ptr())
        fn()
        torch.cuda.synchronize()
        runs.append((pathlib.Path(path).name, fused, fn, R.clone(), []))
base = runs[1][3]
for name, fused, fn, Rv, _t in runs:
    err = float((Rv - base).abs().max() / base.abs().max())
    print(f"{name} fused={fused}: max rel diff of R against the two calls {err:.2e}, plastic {float((dp > 0).double().mean()):.2f}", flush=True)
for rnd in range(5):
    for name, fused, fn, _Rv, times in runs:
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(10):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 10)
for name, fused, _fn, _Rv, times in runs:
    print(json.dumps({"lib": name, "fused": fused, "points": npts, "residual_ms_median": round(statistics.median(times), 4),
                      "all": [round(t, 4) for t in times]}), flush=True)
