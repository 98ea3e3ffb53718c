#!/usr/bin/env python3
"""Does a kernel-calibrated arena block keep its rate when the process goes on allocating? vm_tile (d = 6, 10^7 points)
into the block after each of: more torch allocations, a second Context, freeing things again."""
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
outs = ctx.vm_output_tensors(n, d)
info = outs[0].dxo_block.info
print({k: info[k] for k in ("chosen_kind", "chosen_GBps", "tuned_blocks_per_cu")})
args = (slab.data_ptr(), slab.data_ptr() + n * 48, slab.data_ptr() + n * 96) + tuple(t.data_ptr() for t in outs)


def rate(c=ctx):
    for _ in range(5):
        c.von_mises(prm, d, n, MEM_DEVICE, *args)
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(20):
            c.von_mises(prm, d, n, MEM_DEVICE, *args)
        b.record(stream)
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 20)
    return round(448 * n / statistics.median(ts) / 1e6)


print("fresh block:", rate())
keep = [torch.empty(n * 36, dtype=torch.float64, device=dev) for _ in range(1)]
print("after one more 2.9 GB torch allocation:", rate())
keep += [torch.empty(n * 36, dtype=torch.float64, device=dev) for _ in range(4)]
print("after five:", rate())
c2 = Context(0)
c2.set_stream(stream.cuda_stream)
print("after a second Context:", rate(), " the second context's own launches (no tuned shape: not its block):", rate(c2))
c2.set_option("blocks_per_cu", info["tuned_blocks_per_cu"])
print("second context with blocks_per_cu set:", rate(c2))
clone = outs[0].clone()
print("after cloning C_tang:", rate())
del keep, clone
torch.cuda.empty_cache()
print("after freeing them:", rate())
