#!/usr/bin/env python3
"""How much does physical placement of the output slab change the von Mises kernel's bandwidth?
Allocates K candidate output slabs and input slabs and times the same kernel on each (GPU box)."""
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

n, d, K = 10_000_000, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
g = torch.Generator(device=dev)
g.manual_seed(1)
ins = []
for k in range(K):
    slab = torch.empty(n * 13, dtype=torch.float64, device=dev)
    slab[: n * 6].normal_(0, 3e-3, generator=g)
    slab[n * 6: n * 12].normal_(0, 100.0, generator=g)
    slab[n * 12:].normal_(0, 1e-3, generator=g).abs_()
    ins.append(slab)
outs = [torch.empty(n * 43, dtype=torch.float64, device=dev) for _ in range(K)]


def run(i, o):
    a, b = ins[i], outs[o]
    ctx.von_mises(prm, d, n, MEM_DEVICE, a.data_ptr(), a.data_ptr() + n * 48, a.data_ptr() + n * 96, b.data_ptr(),
                  b.data_ptr() + n * 288, b.data_ptr() + n * 336)


def t(fn, reps=6):
    fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(stream); fn(); b.record(stream)
    torch.cuda.synchronize()
    return statistics.median(x.elapsed_time(y) for x, y in ev)


tiles = n // 128
print("out slab sweep (input slab 0):")
for o in range(K):
    ms = t(lambda: run(0, o))
    pm = t(lambda: ctx.stream_probe(13, 43, tiles, ins[0].data_ptr(), outs[o].data_ptr()))
    print(f"  out {o}: vm {4.48e9 / ms / 1e6:7.1f} GB/s   probe {4.48e9 / pm / 1e6:7.1f} GB/s")
best_o = max(range(K), key=lambda o: 1.0 / t(lambda: run(0, o)))
print("in slab sweep (best out slab", best_o, "):")
for i in range(K):
    ms = t(lambda: run(i, best_o))
    print(f"  in {i}: vm {4.48e9 / ms / 1e6:7.1f} GB/s")

# option sweep on the fastest and the slowest output slab
rate = {o: 1.0 / t(lambda: run(0, o)) for o in range(K)}
fast_o, slow_o = max(rate, key=rate.get), min(rate, key=rate.get)
for name, o in (("fast", fast_o), ("slow", slow_o)):
    for nt in (1, 0):
        for bpc in (0, 4, 8, 16, 32):
            ctx.set_option("nontemporal", nt)
            ctx.set_option("blocks_per_cu", bpc)
            ms = t(lambda: run(0, o), reps=8)
            pm = t(lambda: ctx.stream_probe(13, 43, tiles, ins[0].data_ptr(), outs[o].data_ptr()), reps=8)
            print(f"  {name} slab {o}: nt={nt} bpc={bpc:2d}  vm {4.48e9 / ms / 1e6:7.1f}  probe {4.48e9 / pm / 1e6:7.1f} GB/s")
