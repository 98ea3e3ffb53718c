#!/usr/bin/env python3
"""The constitutive side of the reference's von Mises demo loop (doc/demo/demo_plasticity_von_mises.py:445-456,
:540-566) with every piece of this repository in place, on a mesh built without DOLFINx:

    for each load step:                                          # :540
        for each Newton iteration:                               # SNES, petsc/petsc.py:60
            evaluated_operands = evaluate_operands([sigma])      # deps = eps(Du): device, lazy
            ((C_tang, sigma, dp),) = evaluate_external_operators([J_sigma], evaluated_operands)
                                                                 # one launch: strain + return map + tangent
        p += dp; sigma_n[:] = sigma                              # :564-565

There is no assembly or linear solve here (that is DOLFINx/PETSc territory and out of scope): the displacement
increment of every "Newton iteration" is prescribed, so the script shows the calling sequence and what one
constitutive update costs. Needs an MI355X.   python3 examples/von_mises_load_stepping.py [cells_per_side] [host|resident]

With `resident` the history variables live in a device mirror (make_von_mises(state="resident")): they are uploaded at the
first call, a call sends the dof vector only, and the load-step update gets ONE extra line, `commit_state()`.
"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

from dolfinx_external_operator_amd import (DeviceMesh, QuadratureExternalOperator, evaluate_external_operators,  # noqa: E402
                                           evaluate_operands, make_von_mises)
from tools.synthetic import structured_mesh  # noqa: E402


def main(n_side: int = 100, state: str = "host") -> dict:
    mesh = structured_mesh("triangle", (n_side, n_side), degree=2, distort=0.15, seed=0)   # P2 triangles, 3 qp/cell
    dmesh = DeviceMesh.from_synthetic(mesh)
    n_pts = mesh.num_cells * mesh.nq
    Du = np.zeros(mesh.node_x.shape[0] * 2)                    # displacement increment of the current load step
    sigma_n = np.zeros(n_pts * 4)                               # state: closure-captured, re-read at every call (:347-348)
    p = np.zeros(n_pts)
    deps = dmesh.operand("eps", lambda: Du, lazy=True)          # the operand eps(Du), :225-227
    sigma_op = QuadratureExternalOperator(deps, num_cells=mesh.num_cells, num_points=mesh.nq, value_shape=(4, 4),
                                          external_function=make_von_mises(lambda: sigma_n, lambda: p, state=state), derivatives=(1,))
    x = mesh.node_x
    shape = np.stack([x[:, 0] * (1 + 0.3 * x[:, 1]), -0.3 * x[:, 1] + 0.1 * x[:, 0] ** 2], axis=1).reshape(-1)   # a smooth mode
    report = {"points": n_pts, "state": state, "steps": []}
    for step, load in enumerate([1.5e-3, 3e-3, 4.5e-3, 3e-3]):          # loading ... then unloading
        t0 = time.perf_counter()
        for it in range(3):                                     # stand-in for the Newton iterations of one load step
            Du[:] = shape * (load if step == 0 else load - [1.5e-3, 3e-3, 4.5e-3][step - 1]) * (0.5 + 0.25 * it)
            evaluated = evaluate_operands([sigma_op])
            ((C_tang, sigma, dp),) = evaluate_external_operators([sigma_op], evaluated)
        p += dp                                                 # :564
        sigma_n[:] = sigma                                      # :565
        if state == "resident":
            sigma_op.external_function.commit_state()           # the same update on the device mirror, no transfer
        dt = time.perf_counter() - t0
        report["steps"].append({"load": load, "plastic_fraction": float((dp > 0).mean()), "max_p": float(p.max()),
                                "ms_per_constitutive_update": dt / 3 * 1e3})
        print(f"step {step}: load {load:.1e}  plastic {report['steps'][-1]['plastic_fraction']:.2f}  max p {p.max():.3e}  "
              f"{dt / 3 * 1e3:.2f} ms per update ({n_pts} points)")
    assert np.array_equal(sigma_op.ref_coefficient.x.array, C_tang)   # the operator's coefficient holds the tangent (:441)
    if state == "resident":
        assert sigma_op.external_function.check_state() == 0.0        # the mirror followed the host arrays exactly
    report["final_p"], report["final_sigma_n"] = p.copy(), sigma_n.copy()
    dmesh.close()
    return report


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, sys.argv[2] if len(sys.argv) > 2 else "host")
