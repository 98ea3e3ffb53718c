#!/usr/bin/env python3
"""A/B timing of builds of the fused strain + return-map kernel (dxo_von_mises_field) on ONE mesh in ONE process.
usage: python scripts/exp/vmfield_ab.py [lib.so ...]   (default: the in-tree library + every build_exp/libdxo_*.so)
Q2 hexahedra, 108^3 cells = 1.008e7 points, d = 6; rounds are interleaved so drifts hit every variant alike."""
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
from dolfinx_external_operator_amd import MEM_DEVICE, Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

libs = sys.argv[1:] or [str(L.LIB_PATH)] + sorted(glob.glob(str(ROOT / "dolfinx_external_operator_amd" / "build_exp" / "libdxo_*.so")))
import os  # noqa: E402
n = int(os.environ.get("VMF_N", "108"))
tri = os.environ.get("VMF_CELL", "hex") == "tri"               # VMF_CELL=tri: P2 triangles, 1291^2 x 2 cells
tet = os.environ.get("VMF_CELL", "hex") == "tet"               # VMF_CELL=tet: P2 tetrahedra, 75^3 x 6 cells, 4 points each
no_tangent = os.environ.get("VMF_NOTANGENT", "0") == "1"       # VMF_NOTANGENT=1: the (sigma, dp)-only launch (C_tang = NULL)
m = (structured_mesh("triangle", (1291, 1291), 2, distort=0.2, seed=0) if tri else structured_mesh("tetrahedron", (75,) * 3, 2, distort=0.2, seed=0) if tet
     else structured_mesh("hexahedron", (n, n, n), 2, distort=0.2, seed=0))
dev = torch.device("cuda:0")
npts, d = m.num_cells * m.nq, 4 if tri else 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
rng = np.random.Generator(np.random.PCG64(0))
u = torch.from_numpy(rng.normal(0.0, 3e-3, size=m.node_x.shape[0] * m.gdim)).to(dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
sig = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64) * 100
pp = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
stream = torch.cuda.current_stream()
runs = []
first = None
for path in libs:
    L._lib = L.load_library(path)
    ctx = Context(0)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_option("operand_cell", 0)
    if os.environ.get("VMF_NT") is not None:
        ctx.set_option("nontemporal", int(os.environ["VMF_NT"]))
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    if not runs:
        shared_out = ctx.vm_output_tensors(npts, d)   # ONE block for every variant (calibrated with vm_tile: the fused kernel stores in the same pattern)
    Ct, st, dpt = shared_out
    fn = lambda dm=dm, Ct=Ct, st=st, dpt=dpt: dm.von_mises(prm, u.data_ptr(), sig.data_ptr(), pp.data_ptr(), None if no_tangent else Ct.data_ptr(), st.data_ptr(), dpt.data_ptr(), mem=MEM_DEVICE)  # noqa: E731
    fn()
    torch.cuda.synchronize()
    if first is None:
        first = (Ct.clone(), st.clone())
    else:
        if not (torch.equal(first[1], st) and torch.equal(first[0], Ct)):
            print(f"{path}: results differ from the first library", flush=True)
    runs.append((pathlib.Path(path).name, fn, ctx, dm, (Ct, st, dpt), []))
for rnd in range(5):
    for name, fn, *_rest, times in runs:
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(10):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 10)
for name, *_rest, times in runs:
    print(json.dumps({"lib": name, "points": npts, "fused_ms_median": statistics.median(times), "fused_ms_all": [round(t, 4) for t in times],
                      "placement_GBps": _rest[3][0].dxo_block.info.get("chosen_GBps")}), flush=True)
