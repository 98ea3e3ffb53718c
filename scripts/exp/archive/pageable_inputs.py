#!/usr/bin/env python3
"""Host call (rebuild + resident state) at 10^7 points, d = 6: deps in page-locked memory against deps in an ordinary
NumPy array (what DOLFINx's Expression.eval returns); outputs page-locked in both cases (the factory's recycled pool)."""
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
deps_pin, sigma_n, p, C_tang, sigma, dp = (ctx.pinned_empty(m) for m in (n * d, n * d, n, n * d * d, n * d, n))
blk = min(n, 1_000_000)
reps = -(-n // blk)
deps_pin[:] = np.tile(rng.normal(0.0, 3e-3, size=blk * d), reps)[: n * d]
sigma_n[:] = np.tile(rng.normal(0.0, 100.0, size=blk * d), reps)[: n * d]
p[:] = np.tile(np.abs(rng.normal(0.0, 1e-3, size=blk)), reps)[:n]
deps_page = np.array(deps_pin)            # ordinary (pageable) memory
sn_page, p_page = np.array(sigma_n), np.array(p)
st = ctx.vm_state(d, n)
st.upload(sigma_n, p)
ctx.set_option("vm_host_tangent", 1)


def med(fn):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[1]
    tm = ctx.last_timing()
    return {"ms": round(t * 1e3, 2), "qp_per_s_e8": round(n / t / 1e8, 2), "h2d_ms": round(tm["h2d_ms"], 1), "d2h_ms": round(tm["d2h_ms"], 1)}


print(json.dumps({"resident, deps page-locked": med(lambda: st.call(prm, MEM_HOST, deps_pin, C_tang, sigma, dp))}))
print(json.dumps({"resident, deps pageable": med(lambda: st.call(prm, MEM_HOST, deps_page, C_tang, sigma, dp))}))
print(json.dumps({"rebuild, all inputs page-locked": med(lambda: ctx.von_mises(prm, d, n, MEM_HOST, deps_pin, sigma_n, p, C_tang, sigma, dp))}))
print(json.dumps({"rebuild, all inputs pageable": med(lambda: ctx.von_mises(prm, d, n, MEM_HOST, deps_page, sn_page, p_page, C_tang, sigma, dp))}))
