#!/usr/bin/env python3
"""Where does the host-rebuilt tangent (option vm_host_tangent = 1: only (sigma, dp) cross PCIe, C_tang rebuilt by host threads) start to beat
the copy mode? ms per dxo_von_mises call (NumPy in / out, d = 6 and 4), both modes at several sizes and host-thread counts. Round 6: in the
rebuild mode the device no longer writes a tangent at all. usage: python scripts/exp/rebuild_threshold.py"""
import json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from dolfinx_external_operator_amd import MEM_HOST, Context, VmParams  # noqa: E402

ctx = Context(0)
prm = VmParams(70e3, 0.3, 250.0, 70e3 * 700.0 / (70e3 - 700.0))
rng = np.random.Generator(np.random.PCG64(3))
ctx.set_option("vm_rebuild_min_points", 0)
for d in (6, 4):
    for n in (15_000, 30_000, 60_000, 120_000, 250_000, 500_000):
        deps = rng.normal(0, 3e-3, (n, d)); sn = rng.normal(0, 100.0, (n, d)); p = np.abs(rng.normal(0, 1e-3, n))
        outs = [ctx.pinned_recycled(m) for m in (n * d * d, n * d, n)]
        row = {"d": d, "n": n}
        for label, mode, thr in (("copy", 0, 32), ("rebuild_8", 1, 8), ("rebuild_32", 1, 32)):
            ctx.set_option("vm_host_tangent", mode)
            ctx.set_option("host_threads", thr)
            for _ in range(3):
                ctx.von_mises(prm, d, n, MEM_HOST, deps, sn, p, *outs)
            reps = max(5, min(200, int(2e6 // n)))
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.von_mises(prm, d, n, MEM_HOST, deps, sn, p, *outs)
            row[label] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        print(json.dumps(row), flush=True)
ctx.close()
