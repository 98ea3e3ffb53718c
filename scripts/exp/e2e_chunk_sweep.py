"""The NumPy boundary at 10^7 points (bench.py end_to_end) against the pipeline's chunk size and host thread count: ms per call,
rebuild mode (vm_host_tangent = 1) and resident state, page-locked arrays. usage: python scripts/exp/e2e_chunk_sweep.py [n]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from dolfinx_external_operator_amd import MEM_HOST, Context, VmParams

n, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000), 6
ctx = Context(0)
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
rng = np.random.Generator(np.random.PCG64(7))
bufs = [ctx.pinned_empty(m) for m in (n * d, n * d, n, n * d * d, n * d, n)]
deps, sigma_n, p, C_tang, sigma, dp = bufs
blk = 1_000_000
reps = -(-n // blk)
deps[:] = np.tile(rng.normal(0.0, 3e-3, size=blk * d), reps)[: n * d]
sigma_n[:] = np.tile(rng.normal(0.0, 100.0, size=blk * d), reps)[: n * d]
p[:] = np.tile(np.abs(rng.normal(0.0, 1e-3, size=blk)), reps)[:n]


def med(fn, calls=4):
    ts = []
    for _ in range(calls + 1):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[len(ts[1:]) // 2] * 1e3


ctx.set_option("vm_host_tangent", 1)
base_threads = ctx.get_option("host_threads")
for threads in (base_threads, 16, 64):
    ctx.set_option("host_threads", threads)
    for chunk in (1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21):
        ctx.set_option("host_chunk_points", chunk)
        reb = med(lambda: ctx.von_mises(prm, d, n, MEM_HOST, deps, sigma_n, p, C_tang, sigma, dp))
        st = ctx.vm_state(d, n)
        st.upload(sigma_n, p)
        res = med(lambda: st.call(prm, MEM_HOST, deps, C_tang, sigma, dp))
        st.close()
        print(json.dumps({"host_threads": threads, "host_chunk_points": chunk, "rebuild_ms": round(reb, 2), "resident_ms": round(res, 2),
                          "rebuild_Gqps": round(n / reb / 1e6, 3), "resident_Gqps": round(n / res / 1e6, 3)}), flush=True)
