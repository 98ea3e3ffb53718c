#!/usr/bin/env python3
"""ICNN kernel timing (GPU box). usage: python3 scripts/bench_icnn.py [--n 10000000] [--variant 2] [--precision 0]"""
import argparse, json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np, torch  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--variant", type=int, default=2, help="0 VALU, 1 fp32-input MFMA, 2 split-bf16 MFMA (default)")
ap.add_argument("--precision", type=int, default=0)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--cpu", type=int, default=0, help="also time the NumPy oracle (oracle/icnn_oracle.py) on this many points")
a = ap.parse_args()
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
ctx.set_option("icnn_variant", a.variant)
w = {k.replace("__", "."): v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
g = torch.Generator(device=dev); g.manual_seed(3)
F = torch.randn(a.n, 4, device=dev, dtype=torch.float64, generator=g) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
det = F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]
F[det <= 0.2] = torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
dP = torch.empty(a.n * 16, device=dev, dtype=torch.float64); P = torch.empty(a.n * 4, device=dev, dtype=torch.float64)
run = lambda: ctx.icnn_eval(model, a.precision, a.n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr())
run(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
for x, y in ev:
    x.record(stream); run(); y.record(stream)
torch.cuda.synchronize()
ms = statistics.median(x.elapsed_time(y) for x, y in ev)
# GEMM work of the MFMA kernels: five 64x64x64 GEMMs per 64-point tile = 5 * 2 * 64^3 flop per 64 points (variant 2 issues six
# bf16 products per fp32 product)
mfma_flop = 5 * 2 * 64 ** 3 / 64 * a.n
out = {"case": "ICNN", "variant": a.variant, "precision": a.precision, "n": a.n, "kernel_ms": ms, "qp_per_s": a.n / ms * 1e3,
       "GBps_algorithmic": 192 * a.n / ms / 1e6}
if a.variant == 2 and a.precision == 0:
    out["roofline"] = {"bound": "mfma", "achieved": 6 * mfma_flop / ms / 1e9, "peak": 2516.6, "unit": "TFLOP/s",
                       "frac": 6 * mfma_flop / ms / 1e9 / 2516.6, "fp32_equivalent_TFLOPs": mfma_flop / ms / 1e9,
                       "note": "bf16 MFMA (v_mfma_f32_32x32x16_bf16) dense peak; flop = six bf16 products per fp32 product of the five GEMMs"}
if a.variant == 1 and a.precision == 0:
    out["roofline"] = {"bound": "mfma", "achieved": mfma_flop / ms / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                       "frac": mfma_flop / ms / 1e9 / 157.3,
                       "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32) dense peak; flop = the five 64^3 GEMMs per 64-point tile only"}
if a.cpu:
    import os
    import time
    sys.path.insert(0, str(ROOT))
    from oracle.icnn_oracle import icnn_stress_tangent
    wn = {k: v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
    Fh = F[: a.cpu].cpu().numpy()
    icnn_stress_tangent(Fh[:1000], wn)
    t0 = time.perf_counter(); icnn_stress_tangent(Fh, wn); dt = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": a.cpu / dt, "unit": "qp/s", "cores": len(os.sched_getaffinity(0)), "kind": "port",
                           "sample": f"{a.cpu} points of the same batch, oracle/icnn_oracle.py (NumPy jets, BLAS threads as configured)"}
print(json.dumps(out))
ctx.icnn_destroy(model); ctx.close()
