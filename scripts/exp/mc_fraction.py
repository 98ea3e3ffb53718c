#!/usr/bin/env python3
"""Experiment driver: Mohr-Coulomb call time against the plastic fraction of the batch, per kernel variant (GPU box).
usage: python3 scripts/exp/mc_fraction.py [--n 10000000] [--variants 1,2]"""
import argparse, json, pathlib, statistics, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch  # noqa: E402
from tools.mc_inputs import mc_tracing_inputs  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context, McParams  # noqa: E402
from oracle import load_oracle  # noqa: E402  (input generation only)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--variants", default="1,2")
ap.add_argument("--fractions", default="0,0.05,0.31,1")
a = ap.parse_args()
n = a.n
o = load_oracle()
pool_d, pool_s = mc_tracing_inputs(o, 50_000, seed=2)
ref = o.mohr_coulomb(pool_d, pool_s, nthreads=8)
pl, el = np.flatnonzero(ref[3] > 0), np.flatnonzero(ref[3] <= 0)
dev = torch.device("cuda:0")
prm = McParams(6778.0, 0.25, 3.45, np.pi / 6, np.pi / 6, 26 * np.pi / 180, 0.26 * 3.45 / np.tan(np.pi / 6), 1e-8, 200, 0)
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
rng = np.random.default_rng(1)
Ct = torch.empty(n * 16, dtype=torch.float64, device=dev); s = torch.empty(n * 4, dtype=torch.float64, device=dev)
it = torch.empty(n, dtype=torch.int32, device=dev)
y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
pd, ps = torch.from_numpy(pool_d).to(dev), torch.from_numpy(pool_s).to(dev)
for frac in [float(x) for x in a.fractions.split(",")]:
    take = np.where(rng.random(n) < frac, rng.choice(pl, n), rng.choice(el, n))
    idx = torch.from_numpy(take).to(dev)
    deps, sn = pd[idx].contiguous(), ps[idx].contiguous()
    row = {"n": n, "plastic_fraction": frac}
    for v in [int(x) for x in a.variants.split(",")]:
        ctx.set_option("mc_variant", v)
        run = lambda: ctx.mohr_coulomb(prm, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), Ct.data_ptr(), s.data_ptr(), it.data_ptr(), y.data_ptr(), nr.data_ptr(), dl.data_ptr())
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15: run()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for x, z in ev:
            x.record(stream); run(); z.record(stream)
        torch.cuda.synchronize()
        row[f"v{v}_ms"] = round(statistics.median(x.elapsed_time(z) for x, z in ev), 4)
    print(json.dumps(row), flush=True)
