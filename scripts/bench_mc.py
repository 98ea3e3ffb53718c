#!/usr/bin/env python3
"""Mohr-Coulomb kernel timing on the GPU box (BASELINE config 4: ~10^7 points, tracing distribution).
usage: python3 scripts/bench_mc.py [--n 10000000] [--diag 1] [--launches 5]"""
import argparse
import json
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tools.mc_inputs import mc_pool  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context, McParams  # noqa: E402
from oracle import load_oracle  # noqa: E402  (the --cpu leg only; inputs come from the frozen pool tests/golden/mc_tracing_pool.npz)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--diag", type=int, default=1)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--pool", type=int, default=50_000)
ap.add_argument("--variant", type=int, default=2, help="0 lane = point, 1 classify + Newton kernels, 2 single persistent kernel (default)")
ap.add_argument("--bpc", type=int, default=3)
ap.add_argument("--wps", type=int, default=1)
ap.add_argument("--cpu", type=int, default=0, help="also time the CPU oracle (nested dual numbers, OpenMP) on this many points")
ap.add_argument("--arena", type=int, default=0, help="(no gain measured: the call is fp64-bound, 1.39-1.44 ms either way) 1: C_tang and sigma in an arena block chosen by timing this kernel on the candidates "
                                                    "(dxo_output_alloc_probed); 0: plain torch.empty")
args = ap.parse_args()
n = args.n
o = load_oracle()
pool_d, pool_s = mc_pool()
args.pool = pool_d.shape[0]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(2)
idx = torch.randint(0, args.pool, (n,), generator=g, device=dev)
scale = torch.rand(n, 1, generator=g, device=dev, dtype=torch.float64) * 0.5 + 0.5
deps = (torch.from_numpy(pool_d).to(dev)[idx] * scale).contiguous()
sn = torch.from_numpy(pool_s).to(dev)[idx].contiguous()
it = torch.empty(n, dtype=torch.int32, device=dev)
y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
prm = McParams(6778.0, 0.25, 3.45, np.pi / 6, np.pi / 6, 26 * np.pi / 180, 0.26 * 3.45 / np.tan(np.pi / 6), 1e-8, 200, 0)
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
ctx.set_option("mc_variant", args.variant)
ctx.set_option("mc_blocks_per_cu", args.bpc)
ctx.set_option("mc_waves_per_simd", args.wps)
diag = (it.data_ptr(), y.data_ptr(), nr.data_ptr(), dl.data_ptr()) if args.diag else (None, None, None, None)
placement = None
if args.arena:
    Ct, s = ctx.output_tensors_probed((n * 16, n * 4), lambda ptrs, shape: ctx.mohr_coulomb(prm, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), ptrs[0], ptrs[1], *diag),
                                      bytes_per_launch=(224.0 + (28 if args.diag else 0)) * n)
    placement = {k: Ct.dxo_block.info[k] for k in ("chosen_kind", "chosen_GBps", "candidates", "probe")}
else:
    Ct = torch.empty(n * 16, dtype=torch.float64, device=dev)
    s = torch.empty(n * 4, dtype=torch.float64, device=dev)


def run():
    ctx.mohr_coulomb(prm, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), Ct.data_ptr(), s.data_ptr(), *diag)


run()
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
for a, b in evs:
    a.record(stream)
    run()
    b.record(stream)
torch.cuda.synchronize()
ms = statistics.median(a.elapsed_time(b) for a, b in evs)
bpp = 224 + (28 if args.diag else 0)
out = {"case": "Mohr-Coulomb return map + AD tangent, tracing distribution", "variant": args.variant, "bpc": args.bpc, "wps": args.wps, "n": n, "kernel_ms": ms,
       "qp_per_s": n / ms * 1e3, "GBps_algorithmic": bpp * n / ms / 1e6, "bytes_per_qp": bpp, "output_placement": placement}
if args.diag:
    u, c = torch.unique(it, return_counts=True)
    out["iteration_histogram"] = {int(a): int(b) for a, b in zip(u.tolist(), c.tolist())}
    out["plastic_fraction"] = float((y > 0).double().mean())
    out["max_norm_res_converged"] = float(nr[it < 200].max())
# Not HBM-bound: the binding roof is the fp64 vector pipe; both are reported.
out["roofline"] = {"bound": "hbm", "achieved": bpp * n / ms / 1e6, "peak": 8000.0, "unit": "GB/s",
                   "frac": bpp * n / ms / 1e6 / 8000.0,
                   "note": "fp64-VALU/divergence-bound, see profiles/mc_flop.json and the round's *_mc_pmc.json for the VALU-pipe fraction"}
if args.cpu:
    import os
    import time

    m = args.cpu
    hd, hs = deps[:m].cpu().numpy(), sn[:m].cpu().numpy()
    avail = len(os.sched_getaffinity(0))
    scan = {}
    nt = 1
    while nt <= avail:
        t0 = time.perf_counter()
        o.mohr_coulomb(hd, hs, nthreads=nt)
        scan[nt] = m / (time.perf_counter() - t0)
        if nt > 1 and scan[nt] < 0.7 * max(scan.values()):
            break
        nt *= 2
    best = max(scan, key=scan.get)
    out["cpu_baseline"] = {"value": scan[best], "unit": "qp/s", "cores": best, "kind": "port",
                           "sample": f"{m} points of the same batch, oracle/mc_oracle.cpp (jacfwd through the Newton loop restated "
                                     f"with nested dual numbers), OpenMP, thread scan {sorted(scan)}",
                           "value_1core": scan[1]}
print(json.dumps(out))
