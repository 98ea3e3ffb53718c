#!/usr/bin/env python3
"""The reference's two-call sequence evaluate_operands + evaluate_external_operators (value-side mirrors) on Q2 hexahedra,
108^3 cells x 8 points = 1.0*10^7 points, with every piece of this repository in place: the operand eps(Du) evaluated on the
device inside the constitutive launch (lazy operand), the history variables resident, the results written straight into
the operator's coefficient (outputs=)."""
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import (Context, DeviceMesh, QuadratureExternalOperator, evaluate_external_operators,  # noqa: E402
                                           evaluate_operands, make_von_mises)
from tools.synthetic import structured_mesh  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 108
ctx = Context(0)
m = structured_mesh("hexahedron", (ns, ns, ns), 2, distort=0.2, seed=0)
dm = DeviceMesh.from_synthetic(m, ctx=ctx)
n, d = m.num_cells * m.nq, 6
rng = np.random.Generator(np.random.PCG64(3))
Du = rng.normal(0.0, 1e-3, m.node_x.shape[0] * 3)
sigma_n = np.tile(rng.normal(0.0, 100.0, size=8 * d * 1000), n // 8000 + 1)[: n * d].copy()
p = np.abs(np.tile(rng.normal(0.0, 1e-3, size=8000), n // 8000 + 1)[:n]).copy()
for state, use_out, snap in (("host", False, True), ("host", True, True), ("resident", True, True), ("resident", True, False)):
    deps = dm.operand("eps", lambda: Du, lazy=True, snapshot=snap)
    op = QuadratureExternalOperator(deps, num_cells=m.num_cells, num_points=m.nq, value_shape=(d, d), derivatives=(1,))
    op.external_function = make_von_mises(sigma_n, p, ctx=ctx, state=state, outputs=(op.ref_coefficient, None, None) if use_out else None)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ev = evaluate_operands([op])
        evaluate_external_operators([op], ev)
        ts.append(time.perf_counter() - t0)
    t = sorted(ts[1:])[1]
    print(json.dumps({"case": f"lazy eps(Du) operand, state={state}, outputs={'coefficient' if use_out else 'default'}, snapshot={snap}", "points": n,
                      "ms_per_evaluate_pair": round(t * 1e3, 2), "qp_per_s_e8": round(n / t / 1e8, 2)}), flush=True)
