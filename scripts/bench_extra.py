#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (never the headline `value` of bench.py):
  * von Mises d=6 at BASELINE config 2's 10^6 points and d=4 (the reference demo's plane-strain layout)
  * the PCIe-inclusive host entry point (pageable and pinned NumPy buffers)
  * heat flux kernel bandwidth
Each line printed is one JSON object. Run on the GPU box: python3 scripts/bench_extra.py
"""
import json
import pathlib
import statistics
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, MEM_HOST, Context, VmParams  # noqa: E402

dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
E = 70e3
prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
BPP = {4: 240, 6: 448}


def ev_time(fn, launches=20, warm=3):
    for _ in range(warm):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in evs:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return statistics.mean(a.elapsed_time(b) for a, b in evs)


def vm_case(n, d, label, expand=False):
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    deps = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 3e-3
    deps[:, 3:] *= 2 ** 0.5
    sig = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 100
    p = (torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
    C, s, dp = ctx.vm_output_tensors(n, d)   # kernel-calibrated arena block (dxo_vm_output_alloc); below 1 GiB a plain allocation
    ms = ev_time(lambda: ctx.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sig.data_ptr(), p.data_ptr(),
                                       C.data_ptr(), s.data_ptr(), dp.data_ptr()))
    print(json.dumps({"case": label, "n": n, "d": d, "kernel_ms": ms, "qp_per_s": n / ms * 1e3,
                      "GBps": BPP[d] * n / ms / 1e6}), flush=True)
    if expand:
        # the multi-GPU gather's rebuild of remote tangents from (sigma, dp): read (d+1), write d*d doubles per point
        ms = ev_time(lambda: ctx.vm_expand_tangent(prm, d, n, MEM_DEVICE, s.data_ptr(), dp.data_ptr(), C.data_ptr()))
        print(json.dumps({"case": f"tangent rebuild from (sigma, dp), d={d} (dxo_vm_expand_tangent)", "n": n, "kernel_ms": ms,
                          "qp_per_s": n / ms * 1e3, "GBps": 8 * (d * d + d + 1) * n / ms / 1e6}), flush=True)
        ms = ev_time(lambda: ctx.vm_commit_state(d, n, p.data_ptr(), dp.data_ptr(), sig.data_ptr(), s.data_ptr()))
        print(json.dumps({"case": f"history update p += dp, sigma_n <- sigma, d={d} (dxo_vm_commit_state)", "n": n,
                          "kernel_ms": ms, "GBps": 8 * (3 + 2 * d) * n / ms / 1e6}), flush=True)
    return deps, sig, p


vm_case(1_000_000, 6, "von Mises d=6, config 2 size (448 MB working set, inputs fit the 256 MB Infinity Cache)")
vm_case(10_000_000, 6, "von Mises d=6, 1e7 points", expand=True)
# the kernel is branch-free, so all-elastic and all-plastic batches must time like the mixed one
def vm_branch_case(n, d, scale, label):
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    deps = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 3e-3 * scale
    sig = torch.randn(n, d, generator=g, device=dev, dtype=torch.float64) * 100 * scale
    p = (torch.randn(n, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
    C, s, dp = ctx.vm_output_tensors(n, d)   # kernel-calibrated arena block (dxo_vm_output_alloc); below 1 GiB a plain allocation
    ms = ev_time(lambda: ctx.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sig.data_ptr(), p.data_ptr(),
                                       C.data_ptr(), s.data_ptr(), dp.data_ptr()))
    print(json.dumps({"case": label, "n": n, "d": d, "plastic_fraction": float((dp > 0).double().mean()), "kernel_ms": ms,
                      "GBps": BPP[d] * n / ms / 1e6}), flush=True)


vm_branch_case(10_000_000, 6, 0.05, "von Mises d=6, all-elastic batch")
vm_branch_case(10_000_000, 6, 10.0, "von Mises d=6, all-plastic batch")
vm_case(10_000_000, 4, "von Mises d=4 (reference demo layout), 1e7 points")
vm_case(30_000_000, 4, "von Mises d=4, 3e7 points")

# host entry point, PCIe inclusive
n, d = 1_000_000, 6
rng = np.random.default_rng(0)
hd = rng.normal(0, 3e-3, (n, d)); hs = rng.normal(0, 100, (n, d)); hp = np.abs(rng.normal(0, 1e-3, n))
oC, os_, odp = np.empty(n * d * d), np.empty(n * d), np.empty(n)
for label, bufs in (("pageable", (hd, hs, hp, oC, os_, odp)),):
    ctx.von_mises(prm, d, n, MEM_HOST, *bufs)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.von_mises(prm, d, n, MEM_HOST, *bufs); ts.append(time.perf_counter() - t0)
    print(json.dumps({"case": f"host entry point, {label} NumPy buffers, 1e6 points d=6 (H2D+kernel+D2H)",
                      "qp_per_s": n / statistics.median(ts), **ctx.last_timing()}), flush=True)
pin = [ctx.pinned_empty(a.size) for a in (hd, hs, hp, oC, os_, odp)]
for a, b in zip(pin[:3], (hd, hs, hp)):
    a[:] = b.reshape(-1)
ctx.von_mises(prm, d, n, MEM_HOST, *pin)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ctx.von_mises(prm, d, n, MEM_HOST, *pin); ts.append(time.perf_counter() - t0)
print(json.dumps({"case": "host entry point, pinned (dxo_host_alloc) buffers, 1e6 points d=6",
                  "qp_per_s": n / statistics.median(ts), **ctx.last_timing()}), flush=True)

# heat
n = 50_000_000
T = torch.rand(n, device=dev, dtype=torch.float64) + 0.5
sg = torch.randn(n, 2, device=dev, dtype=torch.float64)
q = torch.empty(n * 2, device=dev, dtype=torch.float64); dT = torch.empty_like(q); ds = torch.empty(n * 4, device=dev, dtype=torch.float64)
ms = ev_time(lambda: ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), q.data_ptr(), dT.data_ptr(), ds.data_ptr()))
print(json.dumps({"case": "heat fused q, dq/dT, dq/dsigma, gdim=2, 5e7 points", "kernel_ms": ms, "qp_per_s": n / ms * 1e3,
                  "GBps": 88 * n / ms / 1e6}), flush=True)
del q, dT, ds
# the same with the outputs in an arena block chosen by timing THIS kernel on the candidates (dxo_output_alloc_probed)
q, dT, ds = ctx.output_tensors_probed((n * 2, n * 2, n * 4), lambda ptrs, shape: ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), *ptrs),
                                      bytes_per_launch=88.0 * n)
ms = ev_time(lambda: ctx.heat(1.0, 1.0, 2, n, MEM_DEVICE, T.data_ptr(), sg.data_ptr(), q.data_ptr(), dT.data_ptr(), ds.data_ptr()))
print(json.dumps({"case": "heat fused, 5e7 points, outputs in a block calibrated with the heat kernel", "kernel_ms": ms, "GBps": 88 * n / ms / 1e6,
                  "chosen_kind": q.dxo_block.info["chosen_kind"], "calibration_GBps": q.dxo_block.info["chosen_GBps"]}), flush=True)
del q, dT, ds
# ICNN (BASELINE config 5): F = I + 0.1 N(0,1), det F > 0.2
w = {k.replace("__", "."): v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
for n, prec, variant, label in ((10_000_000, 0, 2, "fp32 network, split-bf16 MFMA kernel (default)"),
                                (10_000_000, 0, 1, "fp32 network, fp32-input MFMA kernel"),
                                (4_000_000, 0, 0, "fp32 network, VALU kernel"),
                                (1_000_000, 1, 0, "fp64 network (tolerance study)")):
    ctx.set_option("icnn_variant", variant)
    Ft = torch.randn(n, 4, device=dev, dtype=torch.float64) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
    det = Ft[:, 0] * Ft[:, 3] - Ft[:, 1] * Ft[:, 2]
    Ft[det <= 0.2] = torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
    dPt = torch.empty(n * 16, device=dev, dtype=torch.float64); Pt = torch.empty(n * 4, device=dev, dtype=torch.float64)
    ms = ev_time(lambda: ctx.icnn_eval(model, prec, n, MEM_DEVICE, Ft.data_ptr(), dPt.data_ptr(), Pt.data_ptr()), launches=5, warm=1)
    print(json.dumps({"case": f"ICNN stress + tangent, {label}", "n": n, "kernel_ms": ms, "qp_per_s": n / ms * 1e3,
                      "GBps_algorithmic": 192 * n / ms / 1e6, "approx_TFLOPs": 45e3 * n / ms / 1e9}), flush=True)
ctx.icnn_destroy(model)
# analytic Isihara model: same I/O as the network, HBM-bound (192 B per point)
from dolfinx_external_operator_amd import IsiharaParams  # noqa: E402
n = 20_000_000
Ft = torch.randn(n, 4, device=dev, dtype=torch.float64) * 0.05 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
dPt = torch.empty(n * 16, device=dev, dtype=torch.float64); Pt = torch.empty(n * 4, device=dev, dtype=torch.float64)
iprm = IsiharaParams(0.5, 1.0, 1.0, 1.5)
ms = ev_time(lambda: ctx.isihara(iprm, n, MEM_DEVICE, Ft.data_ptr(), dPt.data_ptr(), Pt.data_ptr()))
print(json.dumps({"case": "analytic Isihara stress + tangent (dxo_isihara), fp64, 2e7 points", "n": n, "kernel_ms": ms,
                  "qp_per_s": n / ms * 1e3, "GBps": 192 * n / ms / 1e6}), flush=True)
del dPt, Pt
dPt, Pt = ctx.output_tensors_probed((n * 16, n * 4), lambda ptrs, shape: ctx.isihara(iprm, n, MEM_DEVICE, Ft.data_ptr(), *ptrs), bytes_per_launch=192.0 * n)
ms = ev_time(lambda: ctx.isihara(iprm, n, MEM_DEVICE, Ft.data_ptr(), dPt.data_ptr(), Pt.data_ptr()))
print(json.dumps({"case": "analytic Isihara, 2e7 points, outputs in a block calibrated with the Isihara kernel", "kernel_ms": ms,
                  "GBps": 192 * n / ms / 1e6, "chosen_kind": dPt.dxo_block.info["chosen_kind"], "calibration_GBps": dPt.dxo_block.info["chosen_GBps"]}), flush=True)
del dPt, Pt
ctx.close()
