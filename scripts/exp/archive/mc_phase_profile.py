#!/usr/bin/env python3
"""Experiment driver: where the waves of mc_fused spend their cycles. Needs a build with -DDXO_MC_PROF=1 (scripts/exp/build_variants.sh:
the kernel then overwrites the head of `dlambda` with per-wave cycle counts: classification, refill, Newton pass, passes, total).
usage: DXO_HIP_LIBRARY=.../libdxo_mcprof.so python3 scripts/exp/archive/mc_phase_profile.py"""
import json, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch  # noqa: E402
from tools.mc_inputs import mc_default_params, mc_pool_inputs_device  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
n = 10_000_000
dev = torch.device("cuda:0")
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
prm = mc_default_params()
m = 50_000
pd, ps = mc_pool_inputs_device(torch, torch.device('cuda', ctx.device), m, seed=2)
bufs = lambda k: (torch.empty(k * 16, dtype=torch.float64, device=dev), torch.empty(k * 4, dtype=torch.float64, device=dev), torch.empty(k, dtype=torch.int32, device=dev),
                  torch.empty(k, dtype=torch.float64, device=dev), torch.empty(k, dtype=torch.float64, device=dev), torch.zeros(k, dtype=torch.float64, device=dev))
C, s, it, y, nr, dl = bufs(m)
ctx.set_option("mc_variant", 1)
ctx.mohr_coulomb(prm, m, MEM_DEVICE, pd.data_ptr(), ps.data_ptr(), C.data_ptr(), s.data_ptr(), it.data_ptr(), y.data_ptr(), nr.data_ptr(), dl.data_ptr())
torch.cuda.synchronize()
yh = y.cpu().numpy()
pl, el = np.flatnonzero(yh > 0), np.flatnonzero(yh <= 0)
rng = np.random.default_rng(1)
C, s, it, y, nr, dl = bufs(n)
ctx.set_option("mc_variant", 2)
for frac in (0.0, 0.31, 1.0):
    take = torch.from_numpy(np.where(rng.random(n) < frac, rng.choice(pl, n), rng.choice(el, n))).to(dev)
    deps, sn = pd[take].contiguous(), ps[take].contiguous()
    for _ in range(3):
        ctx.mohr_coulomb(prm, n, MEM_DEVICE, deps.data_ptr(), sn.data_ptr(), C.data_ptr(), s.data_ptr(), it.data_ptr(), y.data_ptr(), nr.data_ptr(), dl.data_ptr())
    torch.cuda.synchronize()
    p = dl[: 2048 * 5].cpu().numpy().reshape(2048, 5)
    tot = p[:, 4].mean()
    print(json.dumps({"plastic_fraction": frac, "cycles_per_wave": tot, "classify_frac": p[:, 0].mean() / tot, "refill_frac": p[:, 1].mean() / tot,
                      "pass_frac": p[:, 2].mean() / tot, "iterations": p[:, 3].mean(), "cycles_per_pass": p[:, 2].sum() / p[:, 3].sum(),
                      "classify_cycles_per_iteration": p[:, 0].sum() / p[:, 3].sum(), "refill_cycles_per_iteration": p[:, 1].sum() / p[:, 3].sum()}))
