"""NumPy restatement of operand evaluation at quadrature points — TEST INFRASTRUCTURE ONLY.

Reference: evaluate_operands (src/dolfinx_external_operator/external_operator.py:386-402) delegates to
`fem.Expression(operand, points).eval(mesh, entities)`; the arithmetic lives in DOLFINx/FFCx (fenics-dolfinx
>=0.10,<0.11, pyproject.toml:13-16), which is not installed here. What that call computes for a Lagrange field is
the textbook push-forward restated below: grad u (x_q) = sum_a u_a (x) (J_q^-T grad_ref phi_a(xi_q)), J_q from the
coordinate element. Operand shapes follow the demos (eps: demo_plasticity_von_mises.py:225-227; F = I + grad u:
demo_hyperelasticity.py:479).

Parity status: UNPINNED against DOLFINx (not executable here; the reference's own test at this boundary,
test/test_operands_evaluation.py:55-66, compares against Expression.eval itself). Pinned instead by known answers:
polynomial fields of the element's degree are represented exactly, so their analytic gradients at the physical
quadrature points are the expected output (tests/test_operand_eval.py).
"""
from __future__ import annotations

import numpy as np

VALUE, GRAD, EPS_MANDEL, DEFGRAD, VALUE_GRAD = 0, 1, 2, 3, 4


def eval_operand(kind, bs, u, dofmap, geom_dofmap, x, phi, dphi, dpsi, cells=None):
    """-> (n_cells, nq, value_size). u: blocked field vector; tables as in dxo_mesh_desc."""
    gdim = dphi.shape[2]
    if cells is None:
        cells = np.arange(dofmap.shape[0])
    cells = np.asarray(cells, dtype=np.int64)
    U = np.asarray(u, dtype=np.float64).reshape(-1, bs)[dofmap[cells]]          # (nc, nd, bs)
    if kind == VALUE:
        return np.einsum("cai,qa->cqi", U, phi)
    X = np.asarray(x)[:, :gdim][geom_dofmap[cells]]                              # (nc, ng, G)
    J = np.einsum("cvj,qvk->cqjk", X, dpsi)                                      # dx_j / dxi_k
    K = np.linalg.inv(J)                                                         # dxi_k / dx_j
    gref = np.einsum("cai,qak->cqik", U, dphi)
    g = np.einsum("cqik,cqkj->cqij", gref, K)                                    # du_i / dx_j
    nc, nq = g.shape[:2]
    if kind == GRAD:
        return g.reshape(nc, nq, bs * gdim)
    if kind == VALUE_GRAD:
        return np.concatenate([np.einsum("cai,qa->cqi", U, phi), g.reshape(nc, nq, bs * gdim)], axis=2)
    if bs != gdim:
        raise ValueError("eps / F need a vector field with bs = gdim")
    r = np.sqrt(2.0) * 0.5
    if kind == EPS_MANDEL:
        if gdim == 2:
            return np.stack([g[..., 0, 0], g[..., 1, 1], np.zeros((nc, nq)), r * (g[..., 0, 1] + g[..., 1, 0])], axis=-1)
        return np.stack([g[..., 0, 0], g[..., 1, 1], g[..., 2, 2], r * (g[..., 0, 1] + g[..., 1, 0]),
                         r * (g[..., 0, 2] + g[..., 2, 0]), r * (g[..., 1, 2] + g[..., 2, 1])], axis=-1)
    if kind == DEFGRAD:
        return (g + np.eye(gdim)).reshape(nc, nq, gdim * gdim)
    raise ValueError(kind)
