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
DIV = 8                                # div u = tr(grad u) (test/test_external_operators_evaluation.py:141)
CAUCHY_GREEN, I1, DETF = 5, 6, 7      # nonlinear operands of F = I + grad u: C = F.T * F, tr(C), det(F) (test/test_operands_evaluation.py:32-36)


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
    if kind == DIV:
        return np.einsum("cqii->cq", g)[..., None]
    F = g + np.eye(gdim)
    if kind == DEFGRAD:
        return F.reshape(nc, nq, gdim * gdim)
    if kind == CAUCHY_GREEN:
        return np.einsum("cqki,cqkj->cqij", F, F).reshape(nc, nq, gdim * gdim)      # F.T * F (:33)
    if kind == I1:
        return np.einsum("cqij,cqij->cq", F, F)[..., None]                          # tr(F.T * F) (:35)
    if kind == DETF:
        return np.linalg.det(F)[..., None]                                          # det(F) (:34)
    raise ValueError(kind)


def _geometry(dofmap, geom_dofmap, x, dphi, dpsi, cells):
    gdim = dphi.shape[2]
    X = np.asarray(x)[:, :gdim][geom_dofmap[cells]]
    J = np.einsum("cvj,qvk->cqjk", X, dpsi)
    return np.linalg.inv(J), np.linalg.det(J)


def operand_adjoint(kind, bs, S, weights, dofmap, geom_dofmap, x, phi, dphi, dpsi, n_nodes, cells=None):
    """Adjoint of eval_operand weighted by the quadrature rule: out[node, i] = sum over cells and points of
    w_q |det J_q| B_q^T S_q — the assembled vector of the form inner(S, operand(v)) dx (internal force for
    kind EPS_MANDEL with S = sigma). S has the operand's shape (n_cells, nq, value_size). Returns (n_nodes*bs,)."""
    gdim = dphi.shape[2]
    if cells is None:
        cells = np.arange(dofmap.shape[0])
    cells = np.asarray(cells, dtype=np.int64)
    K, det = _geometry(dofmap, geom_dofmap, x, dphi, dpsi, cells)
    scale = np.asarray(weights)[None, :] * np.abs(det)                          # (nc, nq)
    gphys = np.einsum("qak,cqkj->cqaj", dphi, K)                                 # d phi_a / d x_j
    S = np.asarray(S, dtype=np.float64).reshape(len(cells), phi.shape[0], -1)
    nc, nq = S.shape[:2]
    vh = np.zeros((nc, nq, bs))
    gh = np.zeros((nc, nq, bs, gdim))
    r = np.sqrt(2.0) * 0.5
    if kind == VALUE:
        vh = S
    elif kind in (GRAD, DEFGRAD):
        gh = S.reshape(nc, nq, bs, gdim)
    elif kind == VALUE_GRAD:
        vh, gh = S[..., :bs], S[..., bs:].reshape(nc, nq, bs, gdim)
    elif kind == EPS_MANDEL:
        if gdim == 2:
            gh[..., 0, 0], gh[..., 1, 1] = S[..., 0], S[..., 1]
            gh[..., 0, 1] = gh[..., 1, 0] = r * S[..., 3]
        else:
            gh[..., 0, 0], gh[..., 1, 1], gh[..., 2, 2] = S[..., 0], S[..., 1], S[..., 2]
            gh[..., 0, 1] = gh[..., 1, 0] = r * S[..., 3]
            gh[..., 0, 2] = gh[..., 2, 0] = r * S[..., 4]
            gh[..., 1, 2] = gh[..., 2, 1] = r * S[..., 5]
    elif kind == DIV:
        for i in range(gdim):
            gh[..., i, i] = S[..., 0]
    else:
        raise ValueError(kind)
    contrib = np.einsum("cq,cqi,qa->cai", scale, vh, phi) + np.einsum("cq,cqij,cqaj->cai", scale, gh, gphys)
    out = np.zeros((n_nodes, bs))
    np.add.at(out, dofmap[cells], contrib)
    return out.reshape(-1)


def tangent_apply(C_tang, v, weights, dofmap, geom_dofmap, x, phi, dphi, dpsi, n_nodes):
    """K v = sum_q w |det J| B^T C_tang B v for the eps / Mandel operand (bs = gdim), without forming K."""
    gdim = dphi.shape[2]
    e = eval_operand(EPS_MANDEL, gdim, v, dofmap, geom_dofmap, x, phi, dphi, dpsi)
    d = e.shape[2]
    t = np.einsum("cqrs,cqs->cqr", np.asarray(C_tang).reshape(e.shape[0], e.shape[1], d, d), e)
    return operand_adjoint(EPS_MANDEL, gdim, t, weights, dofmap, geom_dofmap, x, phi, dphi, dpsi, n_nodes)


def eval_operand_facets(kind, bs, u, dofmap, geom_dofmap, x, phi_f, dphi_f, dpsi_f, entities):
    """Codim-1 entities: (cell, local_facet) pairs as evaluate_operands hands them to Expression.eval
    (src/dolfinx_external_operator/external_operator.py:340, 402; test/test_codim_external_operator.py:76-84).
    Tables per local facet: phi_f (nf, nq, ndofs), dphi_f (nf, nq, ndofs, G), dpsi_f (nf, nq, ngeom, G), tabulated at the
    facet quadrature points mapped into the reference cell. -> (n_entities, nq, value_size)."""
    entities = np.asarray(entities, dtype=np.int64).reshape(-1, 2)
    out = None
    for f in np.unique(entities[:, 1]):
        sel = np.flatnonzero(entities[:, 1] == f)
        part = eval_operand(kind, bs, u, dofmap, geom_dofmap, x, phi_f[f], dphi_f[f], dpsi_f[f], entities[sel, 0])
        if out is None:
            out = np.empty((entities.shape[0],) + part.shape[1:])
        out[sel] = part
    if out is None:
        raise ValueError("empty entity list: value size unknown")
    return out
