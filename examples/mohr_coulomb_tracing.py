#!/usr/bin/env python3
"""Yield-surface tracing of the Mohr-Coulomb demo (doc/demo/demo_plasticity_mohr_coulomb.py:854-957) on the GPU.

50 stress paths in the deviatoric plane p = 0.1, 9 loadings of R = 0.7 each; after every loading the returned stress is
projected back onto the plane (:922-923) and becomes the next state. The history variable lives on the device
(`make_mohr_coulomb(state="resident")`): the host applies the demo's own update to its array and tells the operator
(`commit_state()` is for the unprojected update `sigma_n <- sigma`, :728; the projection is a change of the holder of
another kind, so `state_changed()` is the call here). Prints the demo's "max f" per loading and the distance of the traced
locus from the standard Mohr-Coulomb surface (:933-954).
"""
import pathlib
import sys

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

from dolfinx_external_operator_amd import Context, make_mohr_coulomb  # noqa: E402
from tools.mc_inputs import mc_elastic_matrices, mc_path_increment  # noqa: E402


def trace(ctx, n_angles=50, n_loads=9, R=0.7, p=0.1):
    c, phi = 3.45, 30 * np.pi / 180
    theta = np.linspace(-np.pi / 6 + 1e-5, np.pi / 6 - 1e-5, n_angles)
    tr = np.array([1.0, 1.0, 1.0, 0.0])
    _, S = mc_elastic_matrices()
    deps = mc_path_increment(theta, R) @ S.T                       # :868-871, :903
    sigma_n = np.zeros((n_angles, 4))
    sigma_n[:, :3] = p
    ext = make_mohr_coulomb(sigma_n, ctx=ctx, state="resident")
    rows = []
    for i in range(n_loads):
        _, sigma = ext((1,))(deps.reshape(n_angles, 1, 4))
        niter, yielding, norm_res, _ = ext.last_state
        returned = sigma.reshape(n_angles, 4).copy()
        s = returned - np.outer(returned @ tr / 3.0 - p, tr)        # projection on the same deviatoric plane (:922-923)
        sigma_n[:] = s
        ext.state_changed()                                         # the holder changed by something other than :728
        dev = s - np.outer(s @ tr / 3.0, tr)
        rho = np.sqrt(np.sum(dev * dev, axis=1))                    # sqrt(2 J2)
        rows.append({"load": i, "max_f": float(np.max(yielding)), "max_niter": int(niter.max()), "rho": rho, "sigma": s,
                     "sigma_returned": returned, "yielding": yielding.copy()})
    rho_mc = (np.sqrt(2) * (c * np.cos(phi) + p * np.sin(phi))) / (np.cos(theta) - np.sin(phi) * np.sin(theta) / np.sqrt(3))   # :947-952
    return theta, rows, rho_mc


if __name__ == "__main__":
    theta, rows, rho_mc = trace(Context(0))
    for r in rows:
        print(f"Loading path#{r['load']}  max f: {r['max_f']:.6e}  max inner iterations: {r['max_niter']}")
    rel = (rows[-1]["rho"] - rho_mc) / rho_mc
    print(f"traced locus vs standard Mohr-Coulomb: rho/rho_MC - 1 in [{rel.min():+.4f}, {rel.max():+.4f}] "
          f"(the Abbo-Sloan surface is rounded at the corners and has a tension cut-off)")
