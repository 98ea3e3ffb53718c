"""NumPy restatement of the reference's ICNN hyperelastic operator — TEST INFRASTRUCTURE ONLY.

Reference: doc/demo/demo_hyperelasticity.py
  convexLinear :221-239, ICNN.forward :256-300 (features :263-283, fp32 cast :286, layers :288-299),
  H correction :362-381, compute_stress_local :429-443 (P = grad_F W_NN + F @ H),
  vectorized_stress_and_tangent = vmap(jacfwd(compute_stress_local)) :448, dP_dF_impl :451-456.

The reference differentiates with torch.func (reverse mode for P, forward over reverse for the tangent).
Here the same derivatives are propagated explicitly: every neuron carries its value, its 3 first and 9
second derivatives with respect to the network input x = (K1, K2, K3); the feature map F -> x and its
first/second derivatives are analytic. Precision is the reference's: features and the chain rule in the
dtype of F, the network in fp32 (`.float()`, :286).

Parity status: PINNED — tests/golden/icnn_isihara.npz is produced by executing the reference's own classes
and functions with real torch on the shipped weights (tests/golden/make_golden_icnn.py). Agreement is
limited by fp32 rounding inside the network (different summation order): ~1e-6 of the tangent's scale.
"""
from __future__ import annotations

import numpy as np


def softplus(x):
    """torch.nn.functional.softplus, beta = 1, threshold = 20 (used at :238, :293)."""
    with np.errstate(over="ignore"):
        return np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0)))).astype(x.dtype)


def _sigmoid(x):
    with np.errstate(over="ignore"):
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def features(F):
    """x = (K1, K2, K3) of :263-283 with first and second derivatives w.r.t. F = (F11, F12, F21, F22).
    K depend on F through t = |F|^2 (I1 = t + 1, I2 = t + I3) and D = det F (I3 = D^2, J = |D|)."""
    F = np.asarray(F)
    dt = F.dtype
    n = F.shape[0]
    t = np.sum(F * F, axis=1)
    D = F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]
    aD = np.abs(D)
    sg = np.sign(D)
    m = aD ** (-2.0 / 3.0)          # I3^(-1/3)
    nn = m * m                      # I3^(-2/3)
    K = np.stack([(t + 1.0) * m - 3.0, (t + D * D) * nn - 3.0, (aD - 1.0) ** 2], axis=1).astype(dt)
    # partials in (t, D): k_t, k_D, k_tD, k_DD (k_tt = 0 for all three)
    kt = np.stack([m, nn, np.zeros(n, dtype=dt)], axis=1)
    kD = np.stack([(t + 1.0) * (-2.0 / 3.0) * m / D,
                   2.0 * D * nn + (t + D * D) * (-4.0 / 3.0) * nn / D,
                   2.0 * (aD - 1.0) * sg], axis=1)
    ktD = np.stack([(-2.0 / 3.0) * m / D, (-4.0 / 3.0) * nn / D, np.zeros(n, dtype=dt)], axis=1)
    kDD = np.stack([(t + 1.0) * (10.0 / 9.0) * m / (D * D),
                    -(10.0 / 3.0) * nn + (28.0 / 9.0) * (t + D * D) * nn / (D * D),
                    np.full(n, 2.0, dtype=dt)], axis=1)
    gt = 2.0 * F                                                   # grad t
    gD = np.stack([F[:, 3], -F[:, 2], -F[:, 1], F[:, 0]], axis=1)  # grad det
    Ht = 2.0 * np.eye(4, dtype=dt)
    HD = np.zeros((4, 4), dtype=dt)
    HD[0, 3] = HD[3, 0] = 1.0
    HD[1, 2] = HD[2, 1] = -1.0
    dK = kt[:, :, None] * gt[:, None, :] + kD[:, :, None] * gD[:, None, :]
    d2K = (kt[:, :, None, None] * Ht[None, None] + kD[:, :, None, None] * HD[None, None]
           + ktD[:, :, None, None] * (gt[:, None, :, None] * gD[:, None, None, :] + gD[:, None, :, None] * gt[:, None, None, :])
           + kDD[:, :, None, None] * (gD[:, None, :, None] * gD[:, None, None, :]))
    return K, dK.astype(dt), d2K.astype(dt)


def network_jets(x, w, net_dtype=np.float32):
    """y(x), dy/dx (N,3), d2y/dx2 (N,3,3) of the ICNN (:288-299), propagated layer by layer in `net_dtype`."""
    T = net_dtype
    x = x.astype(T)
    W0, b0 = w["layers__0__weight"].astype(T), w["layers__0__bias"].astype(T)
    z = x @ W0.T + b0                                      # :289
    dz = np.broadcast_to(W0[None], (x.shape[0], 64, 3)).astype(T)
    d2z = np.zeros((x.shape[0], 64, 3, 3), dtype=T)
    for layer in (1, 2):
        Wp = softplus(w[f"layers__{layer}__weights"].astype(T).T)          # (in, out), :238
        S, c = w[f"skip_layers__{layer}__weight"].astype(T), w[f"skip_layers__{layer}__bias"].astype(T)
        a = z @ Wp + (x @ S.T + c)                                         # :291-293
        da = np.einsum("nik,io->nok", dz, Wp) + S[None]
        d2a = np.einsum("nikl,io->nokl", d2z, Wp)
        sp, s1 = softplus(a), _sigmoid(a)
        s1 = np.where(a > 20.0, 1.0, s1).astype(T)
        s2 = np.where(a > 20.0, 0.0, s1 * (1.0 - s1)).astype(T)
        p1 = sp * s1 / T(6.0)                                              # d/da [softplus(a)^2 / 12], :294-295
        p2 = (s1 * s1 + sp * s2) / T(6.0)
        z = sp * sp / T(12.0)
        dz = p1[:, :, None] * da
        d2z = p1[:, :, None, None] * d2a + p2[:, :, None, None] * (da[:, :, :, None] * da[:, :, None, :])
    W3 = softplus(w["layers__3__weights"].astype(T).T)                     # (64, 1)
    S3 = softplus(w["skip_layers__3__weights"].astype(T).T)                # (3, 1)
    y = (z @ W3 + x @ S3)[:, 0]                                            # :299
    dy = np.einsum("nik,io->nk", dz, W3) + S3[:, 0][None]
    d2y = np.einsum("nikl,io->nkl", d2z, W3)
    return y, dy, d2y


def h_correction(w, net_dtype=np.float32):
    """H of :362-381: H_flat = -P_NN(F = I), evaluated like the reference with an fp32 F_0."""
    F0 = np.array([[1.0, 0.0, 0.0, 1.0]], dtype=np.float32)
    K, dK, _ = features(F0)
    _, dy, _ = network_jets(K, w, net_dtype)
    h = -(np.einsum("nk,nki->ni", dy.astype(np.float32), dK)[0]).astype(np.float32)
    H = np.array([[h[0], h[1], 0, 0], [h[2], h[3], 0, 0], [0, 0, h[0], h[1]], [0, 0, h[2], h[3]]], dtype=np.float32)
    return H


def icnn_stress_tangent(F, w, net_dtype=np.float32):
    """dP_dF_impl (:451-456): F (N,4) -> dP (N,4,4) with dP[i][j] = dP_i/dF_j, P (N,4); dtype follows F."""
    F = np.ascontiguousarray(F).reshape(-1, 4)
    dt = F.dtype
    K, dK, d2K = features(F)
    _, dy, d2y = network_jets(K, w, net_dtype)
    dy, d2y = dy.astype(dt), d2y.astype(dt)
    H = h_correction(w, net_dtype).astype(dt)
    P = np.einsum("nk,nki->ni", dy, dK) + F @ H                            # :433-439
    dP = np.einsum("nk,nkij->nij", dy, d2K) + np.einsum("nkl,nki,nlj->nij", d2y, dK, dK) + H.T[None]
    return dP, P


def isihara_stress_tangent(F, c=(0.5, 1.0, 1.0, 1.5)):
    """Analytic Isihara model, demo_hyperelasticity.py:686-703 (UFL): W = c1 K1 + c2 K2 + c3 K1^2 + c4 (J-1)^2 with
    K1 = I1bar - 3, K2 = I2bar - 3 — the same (K1, K2, K3) as `features` for det F > 0 (UFL's J = det F; for
    det F <= 0 the real power J^(-2/3) has no value: NaN). Returns dP (N,4,4), P (N,4) like icnn_stress_tangent.

    Parity status: the reference has this model as a UFL form only, which cannot be executed here (no UFL/FFCx):
    UNPINNED against the reference; pinned against torch.func differentiation of the energy as written at
    :692-700 (tests/golden/make_golden_isihara.py -> tests/golden/isihara_analytic.npz)."""
    F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 4)
    K, dK, d2K = features(F)
    n = F.shape[0]
    dy = np.stack([c[0] + 2.0 * c[2] * K[:, 0], np.full(n, c[1]), np.full(n, c[3])], axis=1)
    d2y = np.zeros((n, 3, 3))
    d2y[:, 0, 0] = 2.0 * c[2]
    P = np.einsum("nk,nki->ni", dy, dK)
    dP = np.einsum("nk,nkij->nij", dy, d2K) + np.einsum("nkl,nki,nlj->nij", d2y, dK, dK)
    bad = (F[:, 0] * F[:, 3] - F[:, 1] * F[:, 2]) <= 0.0
    P[bad] = np.nan
    dP[bad] = np.nan
    return dP, P
