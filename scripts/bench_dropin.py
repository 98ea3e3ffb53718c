#!/usr/bin/env python3
"""What a user of the reference sees: wall time of external_function((1,))(deps) with NumPy operands (GPU box)."""
import json
import pathlib
import statistics
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import Context, make_von_mises  # noqa: E402

ctx = Context(0)
nc, nq, d = 125_000, 8, 6    # BASELINE config 2: 10^6 points
rng = np.random.default_rng(0)
deps = rng.normal(0, 3e-3, (nc, nq, d))
sigma_n = rng.normal(0, 100, nc * nq * d)
p = np.abs(rng.normal(0, 1e-3, nc * nq))
for reuse in (False, True):
    ext = make_von_mises(sigma_n, p, ctx=ctx, reuse_outputs=reuse)
    f = ext((1,))
    f(deps)
    ts = []
    for _ in range(7):
        fresh = deps.copy()   # Expression.eval hands over a fresh ndarray every call (external_operator.py:402)
        t0 = time.perf_counter()
        C, s, dp = f(fresh)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"case": f"drop-in external_function, 1e6 points d=6, reuse_outputs={reuse}", "wall_ms": statistics.median(ts) * 1e3,
                      "qp_per_s": nc * nq / statistics.median(ts), **ctx.last_timing()}))
ctx.close()

# ---- small batches: the reference's demos run on meshes of a few hundred to a few thousand points
# (demo_plasticity_von_mises.py: ~50x... P2 triangles, 3 qp/cell), so the fixed cost of one call matters there
ctx = Context(0)
for nc, nq, d in ((200, 3, 4), (2048, 3, 4), (20_000, 3, 4), (200_000, 3, 4)):
    n = nc * nq
    deps = rng.normal(0, 3e-3, (nc, nq, d))
    sigma_n = rng.normal(0, 100, n * d)
    p = np.abs(rng.normal(0, 1e-3, n))
    f = make_von_mises(sigma_n, p, ctx=ctx)((1,))
    for _ in range(5):
        f(deps)
    ts = []
    for _ in range(200 if n < 100_000 else 30):
        t0 = time.perf_counter()
        f(deps)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"case": f"drop-in call latency, {n} points d={d} (pinned output reuse)", "wall_us_median": statistics.median(ts) * 1e6,
                      "wall_us_min": min(ts) * 1e6, **ctx.last_timing()}))
ctx.close()
