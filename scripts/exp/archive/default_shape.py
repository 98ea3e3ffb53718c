#!/usr/bin/env python3
"""vm_tile (d = 6, 10^7 points) into plain torch.empty allocations: one tile per wave (blocks_per_cu 0) against a
persistent grid of 32 workgroups per CU, the same allocation for both, eight allocations held side by side."""
import statistics
import sys
import pathlib

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

dev = torch.device("cuda:0")
n, d = 10_000_000, 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
slab = torch.empty(n * 13, dtype=torch.float64, device=dev)
slab[:n * 6].normal_(0.0, 3e-3, generator=g)
slab[n * 6:n * 12].normal_(0.0, 100.0, generator=g)
slab[n * 12:].normal_(0.0, 1e-3, generator=g).abs_()


def rate(o):
    args = (slab.data_ptr(), slab.data_ptr() + n * 48, slab.data_ptr() + n * 96, o.data_ptr(), o.data_ptr() + n * 288, o.data_ptr() + n * 336)
    for _ in range(5):
        ctx.von_mises(prm, d, n, MEM_DEVICE, *args)
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(12):
            ctx.von_mises(prm, d, n, MEM_DEVICE, *args)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 12)
    return round(448 * n / statistics.median(ts) / 1e6)


blocks = [torch.empty(n * 43, dtype=torch.float64, device=dev) for _ in range(8)]
for sizes in ((10_000_000,),):
    for o in blocks:
        row = []
        for bpc in (0, 32, 0, 32):
            ctx.set_option("blocks_per_cu", bpc)
            row.append(rate(o))
        print("plain allocation: bpc 0 / 32 / 0 / 32:", row)
