#!/usr/bin/env python3
"""Golden vectors for the analytic Isihara model (demo_hyperelasticity.py:686-703).

The reference states the model as a UFL expression and lets UFL differentiate it (`P = ufl.diff(W_Isihara, F_)`, :703); UFL / FFCx are
not installed here, so the form itself cannot be executed. Two things are done instead, and they must agree:

 (a) SOURCE EXECUTION ON A STAND-IN (round 6). The reference's own statements `F_ = ...` ... `P = ufl.diff(W_Isihara, F_)`
     (:691-703) are pulled out of the demo with `ast`, unmodified, and executed in a namespace where `ufl` is a thin shim backed by
     torch (fp64): tensors are wrapped so that `*` between two matrices is UFL's matrix product, `.T` the transpose, `**` a scalar
     power; `ufl.Identity / grad / variable / det / tr` are the obvious operations on a 2 x 2 matrix and `ufl.diff(W, F_)` is the
     gradient of the scalar W with respect to the variable (torch autograd with create_graph, so that the tangent dP/dF follows by
     differentiating P once more). `ufl.grad(u_UFL)` returns the displacement gradient of the point (F - I). This pins the oracle and
     the HIP kernel to "the reference's source on a stand-in UFL", like the Mohr-Coulomb golden on its stand-in JAX — not to UFL itself.
 (b) INDEPENDENT RESTATEMENT (rounds 1-5): the same energy written term by term in torch and differentiated with torch.func.

Output: tests/golden/isihara_analytic.npz (F, P, dP, W — the values of (a); (b) is asserted equal to 1e-13 of the scale before writing).
Run ONLY in the build container (reads /root/reference).
"""
import ast
import pathlib
import types

import numpy as np
import torch

REF = pathlib.Path("/root/reference/doc/demo/demo_hyperelasticity.py")
OUT = pathlib.Path(__file__).resolve().parent
WANT = ("F_", "C", "J_", "I1", "I2", "I1_bar", "I2_bar", "W_Isihara", "P")


# ------------------------------------------------------------------------------------------------ (a) the UFL stand-in
class T:
    """A UFL expression value: a torch scalar or 2 x 2 matrix with UFL's operator meanings."""

    def __init__(self, v):
        self.v = v

    @property
    def T(self):                                   # noqa: N802  (UFL: A.T)
        return T(self.v.transpose(-1, -2))

    def __mul__(self, o):
        o = o.v if isinstance(o, T) else o
        if torch.is_tensor(o) and self.v.dim() == 2 and o.dim() == 2:
            return T(self.v @ o)                    # UFL: tensor * tensor contracts the last index with the first
        return T(self.v * o)

    def __rmul__(self, o):
        return T(o * self.v)

    def __add__(self, o):
        return T(self.v + (o.v if isinstance(o, T) else o))

    __radd__ = __add__

    def __sub__(self, o):
        return T(self.v - (o.v if isinstance(o, T) else o))

    def __pow__(self, e):
        return T(self.v ** e)


def make_ufl(gradu):
    ufl = types.SimpleNamespace()
    ufl.Identity = lambda d: T(torch.eye(d, dtype=torch.float64))
    ufl.grad = lambda u: T(gradu)                   # the displacement gradient of this point
    ufl.variable = lambda e: e                      # differentiation variable: the wrapped tensor itself (a leaf below)
    ufl.det = lambda A: T(A.v[0, 0] * A.v[1, 1] - A.v[0, 1] * A.v[1, 0])
    ufl.tr = lambda A: T(torch.trace(A.v))
    ufl.diff = lambda W, X: T(torch.autograd.grad(W.v, X.v, create_graph=True)[0])
    return ufl


def extract():
    """The reference's statements, unmodified, in source order (:691-703)."""
    tree = ast.parse(REF.read_text())
    body = [n for n in tree.body if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name)
            and n.targets[0].id in WANT and 680 <= n.lineno <= 710]
    assert [n.targets[0].id for n in body] == list(WANT), [n.targets[0].id for n in body]
    return compile(ast.Module(body=body, type_ignores=[]), str(REF), "exec")


def run_reference(code, Fv):
    """(W, P, dP) of one point by executing the reference's statements. F_ must be a leaf to differentiate with respect to: the shim's
    Identity + grad sum is re-rooted right after the first statement by giving `grad` a leaf gradu and shifting the derivative —
    dW/dF = dW/d(gradu) since F = I + gradu."""
    gradu = (Fv.reshape(2, 2) - torch.eye(2, dtype=torch.float64)).clone().requires_grad_(True)
    ufl = make_ufl(gradu)
    ns = {"ufl": ufl, "u_UFL": None, "d": 2}
    # `ufl.diff(W, F_)` differentiates with respect to F_ = I + gradu, whose graph parent is gradu: d/dF_ = d/dgradu
    ufl.diff = lambda W, X: T(torch.autograd.grad(W.v, gradu, create_graph=True)[0])
    exec(code, ns)
    P = ns["P"].v
    dP = torch.stack([torch.autograd.grad(P.reshape(-1)[k], gradu, retain_graph=True)[0].reshape(-1) for k in range(4)])
    return ns["W_Isihara"].v.detach(), P.detach().reshape(-1), dP.detach()


# ------------------------------------------------------------------------------------------------ (b) the restatement
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
    code = extract()
    W, P, dP = zip(*(run_reference(code, Ft[k]) for k in range(Ft.shape[0])))
    W, P, dP = torch.stack(W), torch.stack(P), torch.stack(dP)
    grad = torch.func.grad(W_isihara)
    P_b = torch.func.vmap(grad)(Ft)
    dP_b = torch.func.vmap(torch.func.jacfwd(grad))(Ft)
    W_b = torch.func.vmap(W_isihara)(Ft)
    for name, a, b in (("W", W, W_b), ("P", P, P_b), ("dP", dP, dP_b)):
        err = float((a - b).abs().max() / b.abs().max())
        print(f"source-on-stand-in vs restatement: {name} max rel diff {err:.2e}")
        assert err <= 1e-13, name
    np.savez(OUT / "isihara_analytic.npz", F=F, P=P.numpy(), dP=dP.numpy(), W=W.numpy())
    print("P(I) =", P[0].numpy(), "W(I) =", float(W[0]), "|dP| max", float(dP.abs().max()))


if __name__ == "__main__":
    main()
