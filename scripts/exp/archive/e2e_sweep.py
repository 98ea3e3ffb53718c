#!/usr/bin/env python3
"""Host (NumPy in / NumPy out) von Mises call at 10^7 points, d = 6: sweep of the rebuild-mode chunk size and the host
thread count, plain and with the history variables resident (dxo_vm_state). One JSON line per configuration."""
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import MEM_HOST, Context, VmParams  # noqa: E402

n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 6
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
ctx = Context(0)
rng = np.random.Generator(np.random.PCG64(7))
bufs = [ctx.pinned_empty(m) for m in (n * d, n * d, n, n * d * d, n * d, n)]
deps, sigma_n, p, C_tang, sigma, dp = bufs
blk = min(n, 1_000_000)
reps = -(-n // blk)
deps[:] = np.tile(rng.normal(0.0, 3e-3, size=blk * d), reps)[: n * d]
sigma_n[:] = np.tile(rng.normal(0.0, 100.0, size=blk * d), reps)[: n * d]
p[:] = np.tile(np.abs(rng.normal(0.0, 1e-3, size=blk)), reps)[:n]
st = ctx.vm_state(d, n)
st.upload(sigma_n, p)
ctx.set_option("vm_host_tangent", 1)
for threads in (16, 32, 64):
    ctx.set_option("host_threads", threads)
    for lg in (16, 17, 18, 19, 20):
        ctx.set_option("vm_rebuild_chunk_points", 1 << lg)
        row = {"host_threads": threads, "chunk": f"2^{lg}"}
        for name, fn in (("rebuild", lambda: ctx.von_mises(prm, d, n, MEM_HOST, deps, sigma_n, p, C_tang, sigma, dp)),
                         ("resident", lambda: st.call(prm, MEM_HOST, deps, C_tang, sigma, dp))):
            ts = []
            for _ in range(4):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            t = sorted(ts[1:])[1]
            tm = ctx.last_timing()
            row[name] = {"ms": round(t * 1e3, 2), "qp_per_s": round(n / t / 1e8, 2), "h2d": round(tm["h2d_ms"], 1), "d2h": round(tm["d2h_ms"], 1)}
        print(json.dumps(row), flush=True)
