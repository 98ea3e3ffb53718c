#!/usr/bin/env python3
"""Experiment driver: ICNN kernel time against batch size for the two MFMA kernels (GPU box)."""
import json, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np, torch  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
w = {k.replace("__", "."): v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
for n in (600, 6144, 50_000, 200_000, 1_000_000, 10_000_000):
    F = torch.randn(n, 4, device=dev, dtype=torch.float64) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
    dP = torch.empty(n * 16, device=dev, dtype=torch.float64); P = torch.empty(n * 4, device=dev, dtype=torch.float64)
    row = {"n": n}
    for v in (0, 1, 2):
        ctx.set_option("icnn_variant", v)
        run = lambda: ctx.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr())
        for _ in range(30): run()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for x, y in ev:
            x.record(stream); run(); y.record(stream)
        torch.cuda.synchronize()
        row[f"v{v}_us"] = round(statistics.median(x.elapsed_time(y) for x, y in ev) * 1e3, 1)
    print(json.dumps(row), flush=True)
