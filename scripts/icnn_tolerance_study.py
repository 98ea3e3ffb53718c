#!/usr/bin/env python3
"""BASELINE config 5: fp64 vs fp32 tolerance study of the ICNN operator on the GPU (run on the GPU box).

Inputs: F = I + 0.1 N(0,1), rejected unless det F > 0.2, seed 3. Compared, for P and dP/dF:
  fp32 network, MFMA kernel   (what the reference computes in: `.float()`, demo_hyperelasticity.py:286)
  fp32 network, VALU kernel
  fp64 network
against the fp64-network NumPy oracle (test infrastructure) and against each other, plus the analytic Isihara model
the network approximates. Errors are max |a - b| / max |b|. Prints one JSON object."""
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import MEM_HOST, Context, IsiharaParams  # noqa: E402
from oracle.icnn_oracle import icnn_stress_tangent, isihara_stress_tangent  # noqa: E402

n = 200_000
rng = np.random.Generator(np.random.PCG64(3))
F = np.empty((0, 4))
while F.shape[0] < n:
    cand = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(n, 4))
    F = np.concatenate([F, cand[(cand[:, 0] * cand[:, 3] - cand[:, 1] * cand[:, 2]) > 0.2]])
F = np.ascontiguousarray(F[:n])
w = dict(np.load(ROOT / "tests" / "golden" / "icnn_isihara_weights.npz"))
ctx = Context(0)
model = ctx.icnn_create({k.replace("__", "."): v for k, v in w.items()})


def run(precision, variant):
    ctx.set_option("icnn_variant", variant)
    dP, P = np.empty((n, 4, 4)), np.empty((n, 4))
    ctx.icnn_eval(model, precision, n, MEM_HOST, F, dP, P)
    return dP, P


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


res = {"fp32_mfma_bf16x3": run(0, 2), "fp32_mfma_f32": run(0, 1), "fp32_valu": run(0, 0), "fp64": run(1, 0)}
m = 20_000                                   # the NumPy oracle is slow: a sample of the batch
o64 = icnn_stress_tangent(F[:m], w, net_dtype=np.float64)
o32 = icnn_stress_tangent(F[:m], w, net_dtype=np.float32)
dPa, Pa = np.empty((n, 4, 4)), np.empty((n, 4))
ctx.isihara(IsiharaParams(0.5, 1.0, 1.0, 1.5), n, MEM_HOST, F, dPa, Pa)
ia = isihara_stress_tangent(F[:m])
out = {"points": n, "oracle_sample": m, "det_F_min": float((F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]).min()), "table": {}}
for name, (dP, P) in res.items():
    out["table"][name] = {
        "P_vs_fp64_oracle": rel(P[:m], o64[1]), "dP_vs_fp64_oracle": rel(dP[:m], o64[0]),
        "P_vs_fp32_oracle": rel(P[:m], o32[1]), "dP_vs_fp32_oracle": rel(dP[:m], o32[0]),
        "P_vs_fp64_kernel": rel(P, res["fp64"][1]), "dP_vs_fp64_kernel": rel(dP, res["fp64"][0]),
        "dP_asymmetry": float(np.abs(dP - dP.transpose(0, 2, 1)).max() / np.abs(dP).max()),
    }
out["analytic_isihara"] = {"kernel_vs_oracle_P": rel(Pa[:m], ia[1]), "kernel_vs_oracle_dP": rel(dPa[:m], ia[0]),
                           "network_fp64_vs_analytic_P_max": rel(res["fp64"][1], Pa),
                           "network_fp64_vs_analytic_P_median": float(np.median(np.abs(res["fp64"][1] - Pa)) / np.median(np.abs(Pa)))}
print(json.dumps(out, indent=1))
ctx.icnn_destroy(model)
ctx.close()
