#!/usr/bin/env python3
"""A/B timing of builds of dxo_von_mises (vm_tile) on the SAME device buffers in ONE process: placement is identical for
every variant (the output block comes from the first library's arena), rounds are interleaved.
usage: python scripts/exp/vmtile_ab.py [lib.so ...]   (default: in-tree library + build_exp/libdxo_*.so)"""
import glob
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import dolfinx_external_operator_amd._lib as L  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

libs = sys.argv[1:] or [str(L.LIB_PATH)] + sorted(glob.glob(str(ROOT / "dolfinx_external_operator_amd" / "build_exp" / "libdxo_*.so")))
dev = torch.device("cuda:0")
n, d = 10_000_000, 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
g = torch.Generator(device=dev)
g.manual_seed(1)
deps = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 3e-3, generator=g)
sig = torch.empty(n, d, dtype=torch.float64, device=dev).normal_(0.0, 100.0, generator=g)
pp = torch.empty(n, dtype=torch.float64, device=dev).normal_(0.0, 1e-3, generator=g).abs_()
stream = torch.cuda.current_stream()
runs, outs, first = [], None, None
for path in libs:
    L._lib = L.load_library(path)
    ctx = Context(0)
    ctx.set_stream(stream.cuda_stream)
    if outs is None:
        outs = ctx.vm_output_tensors(n, d)    # block calibrated with the kernel; every variant then runs the shape below
        print(json.dumps({"placement": outs[0].dxo_block.info}), flush=True)
        shape = outs[0].dxo_block.info["tuned_blocks_per_cu"]
    ctx.set_option("blocks_per_cu", shape)
    fn = lambda ctx=ctx: ctx.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sig.data_ptr(), pp.data_ptr(), *(o.data_ptr() for o in outs))  # noqa: E731
    fn()
    torch.cuda.synchronize()
    if first is None:
        first = (outs[0].clone(), outs[1].clone())
    same = bool(torch.equal(first[0], outs[0]) and torch.equal(first[1], outs[1]))
    runs.append((pathlib.Path(path).name, fn, ctx, same, []))
for rnd in range(6):
    for name, fn, ctx, same, times in runs:
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(20):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 20)
for name, fn, ctx, same, times in runs:
    ms = statistics.median(times)
    print(json.dumps({"lib": name, "ms_median": ms, "GBps": 448 * n / ms / 1e6, "bitwise_equal_to_first": same, "ms_all": [round(t, 4) for t in times]}), flush=True)
