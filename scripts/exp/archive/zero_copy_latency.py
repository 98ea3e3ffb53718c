#!/usr/bin/env python3
"""Call latency of the drop-in factories at the reference's demo sizes, copy path vs zero-copy path (option
host_zero_copy_bytes): python3 scripts/exp/archive/zero_copy_latency.py"""
import json
import pathlib
import statistics
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import Context, make_heat, make_von_mises  # noqa: E402

ctx = Context(0)
ctx.set_option("host_small_bytes", int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20)
rng = np.random.default_rng(0)
for nc, nq, d in ((200, 3, 4), (2048, 3, 4), (6000, 3, 4), (20_000, 3, 4)):
    n = nc * nq
    deps = rng.normal(0, 3e-3, (nc, nq, d))
    sigma_n = rng.normal(0, 100, n * d)
    p = np.abs(rng.normal(0, 1e-3, n))
    f = make_von_mises(sigma_n, p, ctx=ctx, host_tangent="copy")((1,))
    row = {"case": f"von Mises d={d}", "points": n, "bytes_in_out": n * (9 + 21) * 8}
    ref = None
    for zc in (0, 1 << 26, 0, 1 << 26):
        ctx.set_option("host_zero_copy_bytes", zc)
        for _ in range(5):
            out = f(deps)
        if ref is None:
            ref = [a.copy() for a in out]
        else:
            assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out, ref))
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            f(deps)
            ts.append(time.perf_counter() - t0)
        row.setdefault("zero_copy_us" if zc else "copy_us", []).append(round(statistics.median(ts) * 1e6, 1))
    print(json.dumps(row), flush=True)
g = np.load(ROOT / "tests" / "golden" / "heat_c1.npz")
T, sigma = np.ascontiguousarray(g["T"]), np.ascontiguousarray(g["sigma"].reshape(g["T"].shape[0], -1))
fq = make_heat(ctx=ctx)((0, 0))
row = {"case": "heat q, config 1", "points": int(T.size)}
for zc in (0, 1 << 26, 0, 1 << 26):
    ctx.set_option("host_zero_copy_bytes", zc)
    for _ in range(5):
        fq(T, sigma)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        fq(T, sigma)
        ts.append(time.perf_counter() - t0)
    row.setdefault("zero_copy_us" if zc else "copy_us", []).append(round(statistics.median(ts) * 1e6, 1))
print(json.dumps(row))
