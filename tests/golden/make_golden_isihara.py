#!/usr/bin/env python3
"""Golden vectors for the analytic Isihara model (demo_hyperelasticity.py:686-703).

The reference states the model as a UFL expression and lets UFL differentiate it (`P = ufl.diff(W_Isihara, F_)`,
:703); UFL/FFCx are not installed here, so the form cannot be executed. This script writes the SAME energy,
term by term as at :692-700, in torch (fp64) and differentiates it with torch.func: P = grad_F W, dP = jacfwd(P).
It pins the NumPy oracle and the HIP kernel against an independent differentiation of the formula; it is not an
execution of the reference ("parity unpinned vs UFL").

Output: tests/golden/isihara_analytic.npz (F, P, dP, W).
"""
import pathlib

import numpy as np
import torch

OUT = pathlib.Path(__file__).resolve().parent


def W_isihara(Fv):
    F = Fv.reshape(2, 2)
    C = F.T @ F                                  # :692
    J = F[0, 0] * F[1, 1] - F[0, 1] * F[1, 0]    # :693 (2x2 determinant written out: torch.linalg.det's LU double-derivative returns NaN at some points)
    I1 = torch.trace(C) + 1.0                    # :694
    I2 = I1 + J ** 2 - 1.0                       # :695
    I1_bar = J ** (-2.0 / 3.0) * I1              # :698
    I2_bar = J ** (-4.0 / 3.0) * I2              # :699
    return 0.5 * (I1_bar - 3.0) + (I2_bar - 3.0) + (I1_bar - 3.0) ** 2 + 1.5 * (J - 1.0) ** 2   # :700


def main():
    rng = np.random.Generator(np.random.PCG64(3))
    F = np.array([1.0, 0.0, 0.0, 1.0]) + 0.1 * rng.normal(size=(1200, 4))
    F = F[(F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]) > 0.2][:1000]
    F[0] = [1.0, 0.0, 0.0, 1.0]
    F[1] = [1.3, 0.0, 0.0, 1.0 / 1.3]
    F[2] = [1.0, 0.4, 0.0, 1.0]
    F[3] = [0.8, 0.0, 0.0, 0.8]
    F[4] = [1.5, 0.2, -0.1, 1.4]
    Ft = torch.from_numpy(F)
    grad = torch.func.grad(W_isihara)
    P = torch.func.vmap(grad)(Ft)
    dP = torch.func.vmap(torch.func.jacfwd(grad))(Ft)
    W = torch.func.vmap(W_isihara)(Ft)
    np.savez(OUT / "isihara_analytic.npz", F=F, P=P.numpy(), dP=dP.numpy(), W=W.numpy())
    print("P(I) =", P[0].numpy(), "W(I) =", float(W[0]), "|dP| max", float(dP.abs().max()))


if __name__ == "__main__":
    main()
