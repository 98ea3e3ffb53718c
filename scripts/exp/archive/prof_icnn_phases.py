"""Experiment driver (not part of the product): prints per-phase cycle counts of icnn_mfma. It needs an INSTRUMENTED
build of csrc/icnn.hip that accumulates __builtin_readcyclecounter() deltas around the four phases into a device array
and exports `dxo_icnn_prof_dump`; the shipped library has neither. Results of the round-1 run are in DESIGN.md 8."""
import sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from dolfinx_external_operator_amd import MEM_DEVICE, Context
from dolfinx_external_operator_amd._lib import load_library
ROOT = "/root/repo"
ctx = Context(0)
w = {k.replace("__", "."): v for k, v in np.load(ROOT + "/tests/golden/icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
n = 10_000_000
dev = torch.device("cuda:0")
F = torch.randn(n, 4, device=dev, dtype=torch.float64) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
dP = torch.empty(n * 16, device=dev, dtype=torch.float64); P = torch.empty(n * 4, device=dev, dtype=torch.float64)
lib = load_library()
for variant in (1, 2):
    ctx.set_option("icnn_variant", variant)
    ctx.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr()); torch.cuda.synchronize()
    lib.dxo_icnn_prof_dump()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr()); e1.record(); torch.cuda.synchronize()
    print("variant", variant, "ms", e0.elapsed_time(e1)); sys.stdout.flush()
    lib.dxo_icnn_prof_dump()
