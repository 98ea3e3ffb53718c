#!/usr/bin/env python3
"""In-process interleaved A/B of the von Mises kernel variants and the HBM stream probe (GPU box).
usage: python3 scripts/sweep_vm.py [--n 10000000] [--d 6] [--rounds 5] [--launches 10]"""
import argparse
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, VmParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--d", type=int, default=6)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=10)
args = ap.parse_args()
n, d = args.n // 128 * 128, args.d
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
deps = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 3e-3
sig = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 100
p = (torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
C = torch.empty(n * d * d, dtype=torch.float64, device=dev)
s = torch.empty(n * d, dtype=torch.float64, device=dev)
dp = torch.empty(n, dtype=torch.float64, device=dev)
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
bpp = {4: 240, 6: 448}[d]
R, W = {4: (9, 21), 6: (13, 43)}[d]
# probe buffers: tiles of 64 lanes x 16 B chunks; same byte volume as the kernel
n_tiles = n // 128
src = torch.randn(n_tiles * R * 128, generator=g, device=dev, dtype=torch.float64)
dst = torch.empty(n_tiles * W * 128, dtype=torch.float64, device=dev)
ncopy = 4_480_000_000 // 2 // 1024 // 4  # tiles for the (4,4) copy: ~2.24 GB read + 2.24 GB written
csrc = torch.randn(ncopy * 4 * 128, generator=g, device=dev, dtype=torch.float64)
cdst = torch.empty_like(csrc)


def run_vm(variant, nt, bpc):
    ctx.set_option("vm_variant", variant)
    ctx.set_option("nontemporal", nt)
    ctx.set_option("blocks_per_cu", bpc)
    ctx.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sig.data_ptr(), p.data_ptr(), C.data_ptr(), s.data_ptr(), dp.data_ptr())


def run_probe(nt, bpc):
    ctx.set_option("nontemporal", nt)
    ctx.set_option("blocks_per_cu", bpc)
    ctx.stream_probe(R, W, n_tiles, src.data_ptr(), dst.data_ptr())


def run_copy(nt, bpc):
    ctx.set_option("nontemporal", nt)
    ctx.set_option("blocks_per_cu", bpc)
    ctx.stream_probe(4, 4, ncopy, csrc.data_ptr(), cdst.data_ptr())


def run_torch_copy():
    cdst.copy_(csrc)


vm_bytes = bpp * n
probe_bytes = n_tiles * (R + W) * 1024
copy_bytes = ncopy * 8 * 1024
cases = []
for variant, nt, bpc in [(1, 1, 0), (1, 0, 0), (1, 1, 4), (1, 1, 8), (1, 1, 16), (1, 0, 8), (0, 1, 0), (0, 1, 16)]:
    cases.append((f"vm v{variant} nt{nt} bpc{bpc}", lambda v=variant, a=nt, b=bpc: run_vm(v, a, b), vm_bytes))
for nt, bpc in [(1, 0), (0, 0), (1, 8), (1, 16), (0, 16)]:
    cases.append((f"probe({R},{W}) nt{nt} bpc{bpc}", lambda a=nt, b=bpc: run_probe(a, b), probe_bytes))
for nt, bpc in [(1, 0), (0, 0), (1, 16), (0, 16)]:
    cases.append((f"copy(4,4) nt{nt} bpc{bpc}", lambda a=nt, b=bpc: run_copy(a, b), copy_bytes))
cases.append(("torch copy_", run_torch_copy, copy_bytes))

res = {c[0]: [] for c in cases}
for name, fn, nbytes in cases:  # warm-up
    fn()
torch.cuda.synchronize()
for r in range(args.rounds):
    for name, fn, nbytes in cases:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(args.launches):
            fn()
        b.record(stream)
        torch.cuda.synchronize()
        res[name].append(a.elapsed_time(b) / args.launches)
print(f"n={n} d={d} bytes/launch={vm_bytes / 1e9:.3f} GB")
for name, fn, nbytes in cases:
    ms = res[name]
    med, best = statistics.median(ms), min(ms)
    print(f"{name:28s} median {med:8.4f} ms  {nbytes / med / 1e6:8.1f} GB/s   best {nbytes / best / 1e6:8.1f} GB/s")
