#!/usr/bin/env python3
"""ICNN A/B on the GPU box: the round-4 library (scripts/exp/bin/libdxo_hip_r04.so, built from the r04 sources) against the
current one, each in its own child process (DXO_HIP_LIBRARY) on the same fixed-seed batch: sha256 of (dP, P) and kernel time.
usage: python3 scripts/exp/archive/icnn_ab.py [--n 10000000]"""
import argparse, hashlib, json, os, pathlib, statistics, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10_000_000)
ap.add_argument("--child", action="store_true")
ap.add_argument("--launches", type=int, default=12)
a = ap.parse_args()
if not a.child:
    out = {}
    for name, lib in (("r04", ROOT / "scripts/exp/bin/libdxo_hip_r04.so"), ("now", ROOT / "dolfinx_external_operator_amd/libdxo_hip.so")):
        env = dict(os.environ, DXO_HIP_LIBRARY=str(lib))
        r = subprocess.run([sys.executable, __file__, "--child", "--n", str(a.n), "--launches", str(a.launches)], env=env, capture_output=True, text=True)
        if r.returncode:
            print(name, "FAILED", r.stderr[-2000:]); continue
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    if len(out) == 2:
        out["bit_identical"] = out["r04"]["sha"] == out["now"]["sha"]
        out["speedup"] = out["r04"]["ms"] / out["now"]["ms"]
    print(json.dumps(out))
    sys.exit(0)
sys.path.insert(0, str(ROOT))
import numpy as np, torch  # noqa: E402
from dolfinx_external_operator_amd import MEM_DEVICE, Context  # noqa: E402
dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)
w = {k.replace("__", "."): v for k, v in np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz").items()}
model = ctx.icnn_create(w)
g = torch.Generator(device=dev); g.manual_seed(3)
F = torch.randn(a.n, 4, device=dev, dtype=torch.float64, generator=g) * 0.1 + torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
det = F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]
F[det <= 0.2] = torch.tensor([1.0, 0, 0, 1.0], device=dev, dtype=torch.float64)
dP = torch.empty(a.n * 16, device=dev, dtype=torch.float64); P = torch.empty(a.n * 4, device=dev, dtype=torch.float64)
run = lambda: ctx.icnn_eval(model, 0, a.n, MEM_DEVICE, F.data_ptr(), dP.data_ptr(), P.data_ptr())
for _ in range(3):
    run()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.launches)]
for x, y in ev:
    x.record(stream); run(); y.record(stream)
torch.cuda.synchronize()
ts = sorted(x.elapsed_time(y) for x, y in ev)
m = min(a.n, 2_000_000)
h = hashlib.sha256(dP[: m * 16].cpu().numpy().tobytes()); h.update(P[: m * 4].cpu().numpy().tobytes())
h.update(dP[-16 * 4096:].cpu().numpy().tobytes())
print(json.dumps({"ms": statistics.median(ts), "ms_min": ts[0], "ms_max": ts[-1], "sha": h.hexdigest()}))
ctx.icnn_destroy(model); ctx.close()
