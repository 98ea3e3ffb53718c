#!/usr/bin/env python3
"""Experiment driver: Mohr-Coulomb call time against the plastic fraction of the batch, per kernel variant (GPU box).
usage: python3 scripts/exp/archive/mc_fraction.py [--n 10000000] [--variants 1,2]"""
import argparse, json, pathlib, statistics, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch  # noqa: E402
from tools.mc_inputs import mc_default_params, mc_pool_inputs_device  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--variants", default="1,2")
ap.add_argument("--fractions", default="0,0.05,0.31,1")
a = ap.parse_args()
n = a.n
dev = torch.device("cuda:0")
prm = mc_default_params()
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
# a pool of tracing states (advanced by the library's own kernel), split into elastic and plastic by the library's yield value
m = 50_000
pd, ps = mc_pool_inputs_device(torch, torch.device('cuda', ctx.device), m, seed=2)
pC, psig = torch.empty(m * 16, dtype=torch.float64, device=dev), torch.empty(m * 4, dtype=torch.float64, device=dev)
pit = torch.empty(m, dtype=torch.int32, device=dev)
py, pnr, pdl = (torch.empty(m, dtype=torch.float64, device=dev) for _ in range(3))
ctx.mohr_coulomb(prm, m, MEM_DEVICE, pd.data_ptr(), ps.data_ptr(), pC.data_ptr(), psig.data_ptr(), pit.data_ptr(), py.data_ptr(), pnr.data_ptr(), pdl.data_ptr())
torch.cuda.synchronize()
yh = py.cpu().numpy()
pl, el = np.flatnonzero(yh > 0), np.flatnonzero(yh <= 0)
rng = np.random.default_rng(1)
Ct = torch.empty(n * 16, dtype=torch.float64, device=dev); s = torch.empty(n * 4, dtype=torch.float64, device=dev)
it = torch.empty(n, dtype=torch.int32, device=dev)
y, nr, dl = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
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
