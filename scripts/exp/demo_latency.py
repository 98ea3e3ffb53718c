#!/usr/bin/env python3
"""Latency of the drop-in boundary at the reference demos' own sizes (round 6): microseconds per call of make_von_mises at 15 000
points (d = 4, NumPy in / NumPy out) and per pass of make_heat at config 1 (6 144 points), for the forms of the small-batch host path:
DMA copies (host_zero_copy_bytes = 0), the kernel on page-locked host memory in one piece and in pieces. `c_us` is the library's own
wall time of the call (dxo_last_timing.total_ms), the rest of a call is the Python binding. usage: python scripts/exp/demo_latency.py"""
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import Context, make_heat, make_von_mises  # noqa: E402

ctx = Context(0)
rng = np.random.Generator(np.random.PCG64(7))
nc, nq, d = 5000, 3, 4
n = nc * nq
deps = rng.normal(0.0, 3e-3, size=(nc, nq, d))
sigma_n = rng.normal(0.0, 100.0, size=(n, d))
p = np.abs(rng.normal(0.0, 1e-3, size=n))
REPS = 300


def time_vm(**kw):
    f = make_von_mises(sigma_n, p, ctx=ctx, **kw)((1,))
    for _ in range(5):
        r = f(deps)
    c_us = []
    t0 = time.perf_counter()
    for _ in range(REPS):
        r = f(deps)
        c_us.append(ctx.last_timing()["total_ms"] * 1e3)
    us = (time.perf_counter() - t0) / REPS * 1e6
    return round(us, 1), round(float(np.median(c_us)), 1), r


ref = None
rows = []
for name, opts in (("chunked pipeline (round 5: small path ends at 2 MiB)", {"host_small_bytes": 2 << 20}),
                   ("one packed H2D + kernel + one packed D2H", {"host_small_bytes": 8 << 20, "host_zero_copy_bytes": 0}),
                   ("kernel on page-locked host memory, one piece", {"host_small_bytes": 8 << 20, "host_zero_copy_bytes": 8 << 20, "host_zero_copy_piece_bytes": 0}),
                   ("the same in pieces of 1 MiB (default)", {"host_small_bytes": 8 << 20, "host_zero_copy_bytes": 8 << 20, "host_zero_copy_piece_bytes": 1 << 20}),
                   ("pieces of 512 KiB", {"host_small_bytes": 8 << 20, "host_zero_copy_bytes": 8 << 20, "host_zero_copy_piece_bytes": 512 << 10})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    us, c_us, r = time_vm()
    if ref is None:
        ref = [a.copy() for a in r]
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(r, ref))
    us_r, c_us_r, _ = time_vm(reuse_outputs=True)
    rows.append({"form": name, "us_per_call": us, "c_us": c_us, "us_per_call_reuse_outputs": us_r, "bit_identical": same})
    print(json.dumps(rows[-1]), flush=True)
# state arrays page-locked by the caller (ctx.pin): read in place as well
ctx.pin(sigma_n); ctx.pin(p)
us, c_us, r = time_vm()
print(json.dumps({"form": "default + state arrays pinned by the caller (Context.pin)", "us_per_call": us, "c_us": c_us,
                  "bit_identical": all(np.array_equal(a, b, equal_nan=True) for a, b in zip(r, ref))}), flush=True)
# (sigma, dp) over PCIe and the tangent rebuilt by host threads (0.6 instead of 2.5 MB come back; default only from 2^18 points on)
old_min, old_thr = ctx.get_option("vm_rebuild_min_points"), ctx.get_option("host_threads")
ctx.set_option("vm_rebuild_min_points", 0)
for thr in (1, 4, 8):
    ctx.set_option("host_threads", thr)
    us, c_us, r = time_vm(host_tangent="rebuild")
    err = max(float(np.nanmax(np.abs(a - b))) / float(np.nanmax(np.abs(b))) for a, b in zip(r, ref))
    print(json.dumps({"form": f'host_tangent="rebuild", vm_rebuild_min_points = 0, {thr} host threads', "us_per_call": us, "c_us": c_us, "max_rel_err_vs_copy": err}), flush=True)
ctx.set_option("vm_rebuild_min_points", old_min)
ctx.set_option("host_threads", old_thr)
us, c_us, r = time_vm(state="resident")
print(json.dumps({"form": 'state="resident"', "us_per_call": us, "c_us": c_us}), flush=True)

# ---- heat, config 1
g = np.load(ROOT / "tests" / "golden" / "heat_c1.npz")
T, sigma = np.ascontiguousarray(g["T"]), np.ascontiguousarray(g["sigma"].reshape(g["T"].shape[0], -1))
pairs = [(T.copy(), sigma.copy()) for _ in range(REPS)]
for name, kw in (("default (identity fusion + tripwire)", {}), ("fuse_by_identity=False: three launches", {"fuse_by_identity": False})):
    ext = make_heat(ctx=ctx, **kw)
    fns = [ext(dv) for dv in ((0, 0), (1, 0), (0, 1))]
    for f in fns:
        f(T, sigma)
    t0 = time.perf_counter()
    for Tq, sq in pairs:
        for f in fns:
            f(Tq, sq)
    print(json.dumps({"heat_cfg1": name, "us_per_step": round((time.perf_counter() - t0) / REPS * 1e6, 1)}), flush=True)
# the reference's NumPy statements (part2.py:215-261) on the same host
A = B = 1.0
t0 = time.perf_counter()
for Tq, sq in pairs:
    s3 = sq.reshape(Tq.shape[0], -1, 2)
    k = 1.0 / (A + B * Tq)
    q = -(k[..., None] * s3)
    dqdT = (B * k ** 2)[..., None] * s3
    dqds = np.zeros(Tq.shape + (2, 2)); dqds[..., 0, 0] = -k; dqds[..., 1, 1] = -k
    q.reshape(-1), dqdT.reshape(-1), dqds.reshape(-1)
print(json.dumps({"heat_cfg1": "reference NumPy statements", "us_per_step": round((time.perf_counter() - t0) / REPS * 1e6, 1)}), flush=True)
ctx.close()
