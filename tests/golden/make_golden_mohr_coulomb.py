#!/usr/bin/env python3
"""Generate golden vectors for the Mohr-Coulomb return mapping + AD-through-the-loop tangent.

Run ONLY in the build container (needs /root/reference). JAX is not installable here, so the reference's
kernel functions are executed FROM THEIR OWN SOURCE (pulled out of
doc/demo/demo_plasticity_mohr_coulomb.py with `ast`, unmodified) inside a namespace where the names
`jax`, `jnp` and `np` are thin shims backed by torch (fp64) and torch.func forward-mode AD:

    jax.jacfwd(f)            -> one torch.func.jvp per basis direction (nestable: dgdsigma inside drdy
                                inside dsigma_ddeps, exactly the reference's nesting, :391, :462, :555)
    jax.lax.cond             -> Python branch on the primal value (what XLA's select computes per point)
    jax.lax.while_loop       -> Python while; forward-mode tangents flow through every iteration, as
                                they do in jacfwd-through-while_loop
    jnp.linalg.solve / norm, jnp.clip / arcsin / sqrt / vdot / c_ / concatenate -> torch equivalents
    np.array / zeros / linalg.inv -> torch tensors (so `dev @ sigma` works on dual tensors);
    np.sin / cos / tan / sqrt on Python floats -> math

What is pulled from the reference: constants (:110-116), J3 ... dgdsigma (:282-391), lmbda ... drdy
(:405-462), Nitermax/tol, return_mapping, dsigma_ddeps (:469-555).
Inputs (built here with real NumPy): the demo's yield-surface tracing paths (:854-877, :902-929) plus
seeded general states with a shear component. Output: tests/golden/mohr_coulomb.npz

This pins the oracle to "the reference source on a stand-in AD backend", not to JAX itself; DESIGN.md
says so.
"""
import ast
import math
import pathlib
import sys
import time
import types

import numpy as real_np
import torch

REF = pathlib.Path("/root/reference/doc/demo/demo_plasticity_mohr_coulomb.py")
OUT = pathlib.Path(__file__).resolve().parent
ROOT = OUT.parents[1]
sys.path.insert(0, str(ROOT))

torch.set_default_dtype(torch.float64)

WANT_ASSIGN = {"E", "nu", "c", "phi", "psi", "theta_T", "a", "coeff3", "dev", "tr", "dgdsigma", "lmbda", "mu", "C_elas",
               "S_elas", "ZERO_VECTOR", "drdy", "ZERO_SCALAR", "dsigma_ddeps"}
WANT_TUPLE_ASSIGN = {("Nitermax", "tol")}
WANT_FUNCS = {"J3", "J2", "theta", "sign", "coeff1", "coeff2", "C", "B", "A", "K", "a_g", "surface", "f", "g", "deps_p",
              "r_g", "r_f", "r", "return_mapping"}


def extract():
    tree = ast.parse(REF.read_text())
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in WANT_FUNCS:
            body.append(node)
        elif isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, ast.Name) and t.id in WANT_ASSIGN:
                body.append(node)
            elif isinstance(t, ast.Tuple) and tuple(e.id for e in t.elts if isinstance(e, ast.Name)) in WANT_TUPLE_ASSIGN:
                body.append(node)
    names = {n.name for n in body if isinstance(n, ast.FunctionDef)}
    assert names == WANT_FUNCS, WANT_FUNCS - names
    return ast.Module(body=body, type_ignores=[])


# ---------------------------------------------------------------------------------------------- shims
def _is_t(x):
    return isinstance(x, torch.Tensor)


def _scalar_fn(tfn, mfn):
    def fn(x):
        return tfn(x) if _is_t(x) else mfn(x)
    return fn


class NpShim:
    pi = math.pi
    sqrt = staticmethod(_scalar_fn(torch.sqrt, math.sqrt))
    sin = staticmethod(_scalar_fn(torch.sin, math.sin))
    cos = staticmethod(_scalar_fn(torch.cos, math.cos))
    tan = staticmethod(_scalar_fn(torch.tan, math.tan))

    @staticmethod
    def array(obj, dtype=None):
        return torch.tensor(obj, dtype=torch.float64)

    @staticmethod
    def zeros(shape, dtype=None):
        return torch.zeros(shape, dtype=torch.float64)

    linalg = types.SimpleNamespace(inv=torch.linalg.inv)


class _ConcatC:
    def __getitem__(self, key):
        spec, *parts = key
        assert spec == "0,1,-1"
        return torch.cat([p.reshape(-1) for p in parts])


def _logical_and(a, b):
    return bool(a) and bool(b)


class JnpShim:
    vdot = staticmethod(torch.dot)
    sqrt = staticmethod(torch.sqrt)
    arcsin = staticmethod(torch.asin)
    abs = staticmethod(torch.abs)
    cos = staticmethod(torch.cos)
    sin = staticmethod(torch.sin)
    c_ = _ConcatC()
    logical_and = staticmethod(_logical_and)
    linalg = types.SimpleNamespace(norm=torch.linalg.norm, solve=torch.linalg.solve)

    @staticmethod
    def clip(x, lo, hi):
        return torch.clamp(x, lo, hi)

    @staticmethod
    def concatenate(parts):
        return torch.cat([p.reshape(-1) for p in parts])


def _cond(pred, true_fun, false_fun, *operands):
    return true_fun(*operands) if bool(pred) else false_fun(*operands)


def _while_loop(cond_fun, body_fun, init):
    state = init
    while bool(cond_fun(state)):
        state = body_fun(state)
    return state


def _jacfwd(fun, has_aux=False):
    """Forward-mode Jacobian w.r.t. the first argument, one jvp per basis direction."""
    def wrapped(x, *rest):
        cols, aux = [], None
        for i in range(x.numel()):
            e = torch.zeros_like(x)
            e[i] = 1.0
            if has_aux:
                def with_tensor_aux(xx):
                    out, a = fun(xx, *rest)  # JAX carries ints (niter) as arrays; torch.func wants tensors
                    return out, tuple(v if _is_t(v) else torch.tensor(v) for v in a)
                _, jv, aux = torch.func.jvp(with_tensor_aux, (x,), (e,), has_aux=True)
            else:
                _, jv = torch.func.jvp(lambda xx: fun(xx, *rest), (x,), (e,))
            cols.append(jv)
        J = torch.stack(cols, dim=-1)
        return (J, aux) if has_aux else J
    return wrapped


JaxShim = types.SimpleNamespace(jacfwd=_jacfwd, lax=types.SimpleNamespace(cond=_cond, while_loop=_while_loop))


def load_reference_kernel():
    ns = {"np": NpShim, "jnp": JnpShim, "jax": JaxShim, "PETSc": types.SimpleNamespace(ScalarType=None),
          "stress_dim": 4}  # stress_dim = 2 * gdim (:168), gdim = 2
    exec(compile(extract(), str(REF), "exec"), ns)
    return ns


def _plain(x):
    if _is_t(x):
        return x.detach().cpu().numpy().copy()
    return x


def evaluate(ns, deps, sigma_n):
    """dsigma_ddeps (:555) at one point -> (C_tang, sigma, niter, yielding, norm_res, dlambda)."""
    C_tang, (sigma, niter, yielding, norm_res, dlambda) = ns["dsigma_ddeps"](torch.tensor(deps), torch.tensor(sigma_n))
    return (_plain(C_tang), _plain(sigma), int(niter), float(_plain(yielding)), float(_plain(norm_res)),
            float(_plain(dlambda)))


# --------------------------------------------------------------------------------------------- inputs
def tracing_inputs(ns, n_angles, n_loads):
    """The demo's yield-surface tracing (:854-877, :919-929), restated with real NumPy for the inputs; the
    state after each load comes from the reference kernel itself."""
    S_elas = _plain(ns["S_elas"])
    tr = real_np.array([1.0, 1.0, 1.0, 0.0])
    eps, R, p = 0.00001, 0.7, 0.1
    th = real_np.linspace(-real_np.pi / 6 + eps, real_np.pi / 6 - eps, n_angles)
    dsig = real_np.zeros((n_angles, 4))
    dsig[:, 0] = (R / real_np.sqrt(2)) * (real_np.cos(th) + real_np.sin(th) / real_np.sqrt(3))
    dsig[:, 1] = (R / real_np.sqrt(2)) * (-2 * real_np.sin(th) / real_np.sqrt(3))
    dsig[:, 2] = (R / real_np.sqrt(2)) * (real_np.sin(th) / real_np.sqrt(3) - real_np.cos(th))
    sig_n = real_np.zeros_like(dsig)
    sig_n[:, :3] = p
    rows = []
    for load in range(n_loads):
        new = real_np.empty_like(sig_n)
        for j in range(n_angles):
            deps = S_elas @ dsig[j]                      # :903
            out = evaluate(ns, deps, sig_n[j])
            rows.append((deps, sig_n[j].copy(), out, load))
            new[j] = out[1]
        dp = new @ tr / 3.0 - p                          # :922-923 projection on the deviatoric plane
        new -= real_np.outer(dp, tr)
        sig_n[:] = new                                   # :929
        print(f"  tracing load {load}: max f = {max(r[2][3] for r in rows[-n_angles:]):.3e}, "
              f"iters = {sorted(set(r[2][2] for r in rows[-n_angles:]))}", flush=True)
    return rows


def general_inputs(ns, n, seed):
    """Seeded states with a shear component (the tracing paths have sigma_xy = 0): a compressive
    hydrostatic state plus a random deviator, pushed by a random increment. Candidates are pre-screened
    with the C++ oracle so that points needing > 12 iterations are left out (each costs minutes here)."""
    from oracle import load_oracle

    o = load_oracle()
    rng = real_np.random.Generator(real_np.random.PCG64(seed))
    S_elas = _plain(ns["S_elas"])
    rows = []
    while len(rows) < n:
        pbar = -rng.uniform(0.0, 4.0)
        devi = rng.normal(0, 0.6, 4)
        devi[:3] -= devi[:3].mean()
        sig_n = real_np.array([pbar, pbar, pbar, 0.0]) + devi
        if o.mc_surface(sig_n[None])[0][0] > 0:   # start inside the yield surface
            continue
        dsig = rng.normal(0, 1.2, 4) * rng.choice([0.2, 1.0, 2.5])
        deps = S_elas @ dsig
        _, _, it, *_ = o.mohr_coulomb(deps[None], sig_n[None])
        if it[0] > 12:
            continue
        rows.append((deps, sig_n, evaluate(ns, deps, sig_n), -1))
    return rows


def main():
    t0 = time.time()
    ns = load_reference_kernel()
    params = {k: float(ns[k]) for k in ("E", "nu", "c", "phi", "psi", "theta_T", "a")}
    params["tol"] = float(ns["tol"])
    params["nitermax"] = int(ns["Nitermax"])
    print("reference constants:", params)
    rows = tracing_inputs(ns, n_angles=12, n_loads=9)
    rows += general_inputs(ns, n=60, seed=7)
    # special points
    Cel = _plain(ns["C_elas"])
    zero = real_np.zeros(4)
    rows.append((zero.copy(), real_np.array([0.1, 0.1, 0.1, 0.0]), evaluate(ns, zero, real_np.array([0.1, 0.1, 0.1, 0.0])), -2))  # deps == 0 -> 0 iterations, C_tang = 0 (:500-505)
    small = real_np.array([1e-6, -2e-6, 5e-7, 3e-7])
    rows.append((small, real_np.array([-1.0, -1.2, -0.8, 0.1]), evaluate(ns, small, real_np.array([-1.0, -1.2, -0.8, 0.1])), -3))  # elastic -> C_elas
    deps = real_np.stack([r[0] for r in rows])
    sigma_n = real_np.stack([r[1] for r in rows])
    C_tang = real_np.stack([r[2][0] for r in rows])
    sigma = real_np.stack([r[2][1] for r in rows])
    niter = real_np.array([r[2][2] for r in rows], dtype=real_np.int32)
    yielding = real_np.array([r[2][3] for r in rows])
    norm_res = real_np.array([r[2][4] for r in rows])
    dlambda = real_np.array([r[2][5] for r in rows])
    tag = real_np.array([r[3] for r in rows], dtype=real_np.int32)
    real_np.savez(OUT / "mohr_coulomb.npz", deps=deps, sigma_n=sigma_n, C_tang=C_tang, sigma=sigma, niter=niter,
                  yielding=yielding, norm_res=norm_res, dlambda=dlambda, tag=tag, C_elas=Cel,
                  **{f"prm_{k}": v for k, v in params.items()})
    print(f"saved {len(rows)} points in {time.time() - t0:.0f} s; iteration histogram:",
          dict(zip(*real_np.unique(niter, return_counts=True))))


if __name__ == "__main__":
    main()
