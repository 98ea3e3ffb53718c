#!/usr/bin/env python3
"""ICNN bf16x3 kernel time against the start skew of the second wave of every SIMD (ctx option icnn_skew, x 1024 cycles)."""
import json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
n = 10_000_000
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
w = {k.replace("__", "."): v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
g = torch.Generator(device=dev); g.manual_seed(3)
F = torch.randn(n, 4, device=dev, dtype=torch.float64, generator=g) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
det = F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]
F[det <= 0.2] = torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
dP = torch.empty(n * 16, device=dev, dtype=torch.float64); P = torch.empty(n * 4, device=dev, dtype=torch.float64)
run = lambda: ctx.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr())
for mask, skew in [(m, k) for m in (0xF0, 0xAA, 0xCC) for k in (0, 4, 8, 12, 16)]:
    ctx.set_option("icnn_skew", skew | (mask << 8))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for x, y in ev:
        x.record(stream); run(); y.record(stream)
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in ev)
    print(json.dumps({"mask": hex(mask), "skew_x1024_cycles": skew, "ms_median": round(statistics.median(ts), 4), "ms_min": round(ts[0], 4)}), flush=True)
ctx.icnn_destroy(model); ctx.close()
