#!/usr/bin/env python3
"""Experiment: do two Mohr-Coulomb calls on two streams overlap (HBM-bound classify of one with the FP64-bound Newton
of the other)?  Compares one call on n points with two concurrent calls on n/2 points each and with 2k calls on n/2k."""
import json
import pathlib
import statistics
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conftest import mc_tracing_inputs  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context, McParams  # noqa: E402
from oracle import load_oracle  # noqa: E402

n = 10_000_000
o = load_oracle()
pool_d, pool_s = mc_tracing_inputs(o, 50_000, seed=2)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(2)
idx = torch.randint(0, 50_000, (n,), generator=g, device=dev)
scale = torch.rand(n, 1, generator=g, device=dev, dtype=torch.float64) * 0.5 + 0.5
deps = (torch.from_numpy(pool_d).to(dev)[idx] * scale).contiguous()
sn = torch.from_numpy(pool_s).to(dev)[idx].contiguous()
Ct = torch.empty(n * 16, dtype=torch.float64, device=dev); s = torch.empty(n * 4, dtype=torch.float64, device=dev)
prm = McParams(6778.0, 0.25, 3.45, np.pi / 6, np.pi / 6, 26 * np.pi / 180, 0.26 * 3.45 / np.tan(np.pi / 6), 1e-8, 200, 0)
for wps in (1, 2):
    for parts in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(min(parts, 2))]
        ctxs = [Context(0) for _ in streams]
        for c, st in zip(ctxs, streams):
            c.set_stream(st.cuda_stream)
            c.set_option("mc_waves_per_simd", wps)
        m = n // parts

        def run():
            for k in range(parts):
                c = ctxs[k % len(ctxs)]
                c.mohr_coulomb(prm, m, MEM_DEVICE, deps.data_ptr() + k * m * 32, sn.data_ptr() + k * m * 32,
                               Ct.data_ptr() + k * m * 128, s.data_ptr() + k * m * 32, None, None, None, None)

        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(json.dumps({"waves_per_simd": wps, "parts": parts, "streams": len(streams), "ms": statistics.median(ts) * 1e3}), flush=True)
        for c in ctxs:
            c.close()
