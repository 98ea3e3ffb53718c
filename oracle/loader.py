"""ctypes loader for liboracle (oracle/dxo_oracle.c [+ mc_oracle.cpp]). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
LIB = HERE / "libdxo_oracle.so"


def build_oracle(force: bool = False) -> pathlib.Path:
    srcs = list(HERE.glob("*.c")) + list(HERE.glob("*.cpp")) + list(HERE.glob("*.hpp")) + [HERE / "Makefile"]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        res = subprocess.run(["make", "-C", str(HERE), "-s", "-B" if force else "-s"], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + res.stdout)
    return LIB


MC_DEFAULTS = dict(E=6778.0, nu=0.25, c=3.45, phi=30 * np.pi / 180, psi=30 * np.pi / 180, theta_T=26 * np.pi / 180,
                   a=None, tol=1e-8, nitermax=200)  # demo_plasticity_mohr_coulomb.py:110-116, 469


class _McParams(C.Structure):
    _fields_ = [("E", C.c_double), ("nu", C.c_double), ("c", C.c_double), ("phi", C.c_double), ("psi", C.c_double),
                ("theta_T", C.c_double), ("a", C.c_double), ("tol", C.c_double), ("nitermax", C.c_int32),
                ("_pad", C.c_int32)]


def mc_params(**kw) -> _McParams:
    d = dict(MC_DEFAULTS)
    d.update(kw)
    if d["a"] is None:
        d["a"] = 0.26 * d["c"] / np.tan(d["phi"])  # :116
    return _McParams(d["E"], d["nu"], d["c"], d["phi"], d["psi"], d["theta_T"], d["a"], d["tol"], int(d["nitermax"]), 0)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _IcnnWeights(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("W0", "b0", "W1", "S1", "c1", "W2", "S2", "c2", "W3", "S3")]


class _OracleMesh(C.Structure):
    _fields_ = [("gdim", C.c_int), ("nd", C.c_int), ("ng", C.c_int), ("nq", C.c_int), ("x_cols", C.c_int), ("dofmap", C.c_void_p),
                ("geom_dofmap", C.c_void_p), ("x", C.c_void_p), ("dphi", C.c_void_p), ("dpsi", C.c_void_p), ("w", C.c_void_p)]


_ICNN_KEYS = ("layers__0__weight", "layers__0__bias", "layers__1__weights", "skip_layers__1__weight", "skip_layers__1__bias",
              "layers__2__weights", "skip_layers__2__weight", "skip_layers__2__bias", "layers__3__weights", "skip_layers__3__weights")


class OracleLib:
    def __init__(self, path=LIB):
        self.lib = C.CDLL(str(path))
        P = C.c_void_p
        self.lib.oracle_von_mises.restype = C.c_int
        self.lib.oracle_von_mises.argtypes = [P, C.c_int, C.c_int64, P, P, P, P, P, P, C.c_int]
        self.lib.oracle_conductivity.restype = C.c_int
        self.lib.oracle_conductivity.argtypes = [C.c_double, C.c_double, C.c_int64, P, P, P]
        self.lib.oracle_heat.restype = C.c_int
        self.lib.oracle_heat.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int64, P, P, P, P, P, C.c_int]
        self.lib.oracle_max_threads.restype = C.c_int
        if hasattr(self.lib, "oracle_mohr_coulomb"):
            self.lib.oracle_mohr_coulomb.restype = C.c_int
            self.lib.oracle_mohr_coulomb.argtypes = [P, C.c_int64] + [P] * 8 + [C.c_int]
            self.lib.oracle_mohr_coulomb_sigma.restype = C.c_int
            self.lib.oracle_mohr_coulomb_sigma.argtypes = [P, C.c_int64] + [P] * 7 + [C.c_int]
            self.lib.oracle_mohr_coulomb_ld.restype = C.c_int
            self.lib.oracle_mohr_coulomb_ld.argtypes = [P, C.c_int64] + [P] * 5 + [C.c_int]
            self.lib.oracle_mc_surface.restype = C.c_int
            self.lib.oracle_mc_surface.argtypes = [P, C.c_int64, P, P, P, P]

    def icnn(self, F, weights, *, nthreads=1):
        """oracle/icnn_oracle_c.c: F (N,4) fp64 -> dP (N,4,4), P (N,4), H (4,4) fp32; `weights`: the raw state_dict arrays
        (tests/golden/icnn_isihara_weights.npz, keys with `__` for `.`). The compiled, threaded form of icnn_oracle.py."""
        self.lib.oracle_icnn.restype = C.c_int
        self.lib.oracle_icnn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        keep = [np.ascontiguousarray(weights[k], dtype=np.float32) for k in _ICNN_KEYS]
        w = _IcnnWeights(*[a.ctypes.data for a in keep])
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 4)
        n = F.shape[0]
        dP, P, H = np.empty((n, 4, 4)), np.empty((n, 4)), np.empty((4, 4), dtype=np.float32)
        rc = self.lib.oracle_icnn(C.byref(w), n, _dp(F), _dp(dP), _dp(P), _dp(H), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_icnn rc={rc}")
        return dP, P, H

    def isihara(self, F, c=(0.5, 1.0, 1.0, 1.5), *, nthreads=1):
        """oracle/icnn_oracle_c.c::oracle_isihara: the compiled, threaded form of icnn_oracle.isihara_stress_tangent -> dP (N,4,4), P (N,4)."""
        self.lib.oracle_isihara.restype = C.c_int
        self.lib.oracle_isihara.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, 4)
        cc = np.asarray(c, dtype=np.float64)
        n = F.shape[0]
        dP, P = np.empty((n, 4, 4)), np.empty((n, 4))
        rc = self.lib.oracle_isihara(_dp(cc), n, _dp(F), _dp(dP), _dp(P), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_isihara rc={rc}")
        return dP, P

    # ---- oracle/operand_oracle_c.c: the consumer-side steps for the eps / Mandel operand, compiled and threaded
    def _mesh(self, m, cells, with_weights):
        """m: a tools.synthetic mesh (or anything with dofmap, geom_dofmap, x | node_x, dphi, dpsi, weights)."""
        keep = {"dofmap": np.ascontiguousarray(m.dofmap[:cells], dtype=np.int32), "geom": np.ascontiguousarray(m.geom_dofmap[:cells], dtype=np.int32),
                "x": np.ascontiguousarray(m.x, dtype=np.float64), "dphi": np.ascontiguousarray(m.dphi, dtype=np.float64),
                "dpsi": np.ascontiguousarray(m.dpsi, dtype=np.float64), "w": np.ascontiguousarray(m.weights, dtype=np.float64)}
        nq, nd, G = keep["dphi"].shape
        desc = _OracleMesh(G, nd, keep["dpsi"].shape[1], nq, keep["x"].shape[1], keep["dofmap"].ctypes.data, keep["geom"].ctypes.data,
                           keep["x"].ctypes.data, keep["dphi"].ctypes.data, keep["dpsi"].ctypes.data, keep["w"].ctypes.data if with_weights else None)
        return desc, keep

    def operand_eps(self, m, u, *, cells=None, nthreads=1):
        """eps(u) in Mandel form at the quadrature points of the first `cells` cells -> (cells, nq, d)."""
        nc = m.dofmap.shape[0] if cells is None else int(cells)
        desc, keep = self._mesh(m, nc, False)
        G = desc.gdim
        u = np.ascontiguousarray(u, dtype=np.float64)
        e = np.empty((nc, desc.nq, 4 if G == 2 else 6))
        self.lib.oracle_operand_eps.restype = C.c_int
        self.lib.oracle_operand_eps.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        rc = self.lib.oracle_operand_eps(C.byref(desc), nc, _dp(u), _dp(e), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_operand_eps rc={rc}")
        return e

    def operand_eps_adjoint(self, m, S, n_nodes, *, cells=None, nthreads=1):
        """sum_q w |det J| B^T S over the first `cells` cells -> (n_nodes * gdim,)."""
        nc = m.dofmap.shape[0] if cells is None else int(cells)
        desc, keep = self._mesh(m, nc, True)
        S = np.ascontiguousarray(S, dtype=np.float64)
        out = np.zeros(n_nodes * desc.gdim)
        self.lib.oracle_operand_eps_adjoint.restype = C.c_int
        self.lib.oracle_operand_eps_adjoint.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        rc = self.lib.oracle_operand_eps_adjoint(C.byref(desc), nc, _dp(S), _dp(out), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_operand_eps_adjoint rc={rc}")
        return out

    def tangent_apply(self, m, C_tang, v, n_nodes, *, cells=None, nthreads=1):
        """K v over the first `cells` cells, K never formed -> (n_nodes * gdim,)."""
        nc = m.dofmap.shape[0] if cells is None else int(cells)
        desc, keep = self._mesh(m, nc, True)
        C_tang = np.ascontiguousarray(C_tang, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros(n_nodes * desc.gdim)
        self.lib.oracle_tangent_apply.restype = C.c_int
        self.lib.oracle_tangent_apply.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        rc = self.lib.oracle_tangent_apply(C.byref(desc), nc, _dp(C_tang), _dp(v), _dp(out), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_tangent_apply rc={rc}")
        return out

    def max_threads(self) -> int:
        return int(self.lib.oracle_max_threads())

    def von_mises(self, deps, sigma_n, p, *, E=70e3, nu=0.3, sigma_0=250.0, H=None, nthreads=1, out=None):
        """deps (..., d), sigma_n (..., d), p (...) -> C_tang (N, d, d), sigma (N, d), dp (N,).
        `out=(C_tang, sigma, dp)` reuses caller buffers (timing runs: keeps page faults out of the loop)."""
        if H is None:
            Et = E / 100.0
            H = E * Et / (E - Et)
        d = deps.shape[-1]
        deps = np.ascontiguousarray(deps, dtype=np.float64).reshape(-1, d)
        sigma_n = np.ascontiguousarray(sigma_n, dtype=np.float64).reshape(-1, d)
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        n = deps.shape[0]
        prm = np.array([E, nu, sigma_0, H], dtype=np.float64)
        if out is None:
            C_tang = np.empty((n, d, d))
            sigma = np.empty((n, d))
            dp = np.empty(n)
        else:
            C_tang, sigma, dp = out
            assert C_tang.size == n * d * d and sigma.size == n * d and dp.size == n
        rc = self.lib.oracle_von_mises(_dp(prm), d, n, _dp(deps), _dp(sigma_n), _dp(p), _dp(C_tang), _dp(sigma),
                                       _dp(dp), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_von_mises rc={rc}")
        return C_tang, sigma, dp

    def heat(self, T, sigma, *, A=1.0, B=1.0, gdim=2, nthreads=1):
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1)
        n = T.size
        sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(n, gdim)
        q = np.empty((n, gdim))
        dqdT = np.empty((n, gdim))
        dqds = np.empty((n, gdim, gdim))
        rc = self.lib.oracle_heat(A, B, gdim, n, _dp(T), _dp(sigma), _dp(q), _dp(dqdT), _dp(dqds), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_heat rc={rc}")
        return q, dqdT, dqds


    def conductivity(self, T, *, A=1.0, B=1.0):
        """k = 1 / (A + B T), dk/dT = -B k^2 (demo_nonlinear_heat_equation_part1.py:251-271); flat arrays."""
        T = np.ascontiguousarray(T, dtype=np.float64).reshape(-1)
        k, dk = np.empty(T.size), np.empty(T.size)
        rc = self.lib.oracle_conductivity(A, B, T.size, _dp(T), _dp(k), _dp(dk))
        if rc != 0:
            raise ValueError(f"oracle_conductivity rc={rc}")
        return k, dk

    def mohr_coulomb(self, deps, sigma_n, *, nthreads=1, tangent=True, **params):
        """deps (N,4), sigma_n (N,4) -> C_tang (N,4,4) [None if tangent=False], sigma (N,4), niter (N,) int32,
        yielding (N,), norm_res (N,), dlambda (N,)"""
        prm = mc_params(**params)
        deps = np.ascontiguousarray(deps, dtype=np.float64).reshape(-1, 4)
        sigma_n = np.ascontiguousarray(sigma_n, dtype=np.float64).reshape(-1, 4)
        n = deps.shape[0]
        sigma = np.empty((n, 4)); niter = np.empty(n, dtype=np.int32)
        yielding = np.empty(n); norm_res = np.empty(n); dlambda = np.empty(n)
        if tangent:
            C_tang = np.empty((n, 4, 4))
            rc = self.lib.oracle_mohr_coulomb(C.byref(prm), n, _dp(deps), _dp(sigma_n), _dp(C_tang), _dp(sigma),
                                              _dp(niter), _dp(yielding), _dp(norm_res), _dp(dlambda), int(nthreads))
        else:
            C_tang = None
            rc = self.lib.oracle_mohr_coulomb_sigma(C.byref(prm), n, _dp(deps), _dp(sigma_n), _dp(sigma), _dp(niter),
                                                    _dp(yielding), _dp(norm_res), _dp(dlambda), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_mohr_coulomb rc={rc}")
        return C_tang, sigma, niter, yielding, norm_res, dlambda

    def mohr_coulomb_long_double(self, deps, sigma_n, *, nthreads=1, **params):
        """The same return map + AD tangent with 80-bit long double under the dual numbers (an accuracy referee, not the
        reference's arithmetic): C_tang (N,4,4), sigma (N,4), niter (N,)."""
        prm = mc_params(**params)
        deps = np.ascontiguousarray(deps, dtype=np.float64).reshape(-1, 4)
        sigma_n = np.ascontiguousarray(sigma_n, dtype=np.float64).reshape(-1, 4)
        n = deps.shape[0]
        C_tang, sigma, niter = np.empty((n, 4, 4)), np.empty((n, 4)), np.empty(n, dtype=np.int32)
        rc = self.lib.oracle_mohr_coulomb_ld(C.byref(prm), n, _dp(deps), _dp(sigma_n), _dp(C_tang), _dp(sigma), _dp(niter), int(nthreads))
        if rc != 0:
            raise ValueError(f"oracle_mohr_coulomb_ld rc={rc}")
        return C_tang, sigma, niter

    def mc_surface(self, sigma, **params):
        """f(sigma), g(sigma), dg/dsigma for sigma (N,4)."""
        prm = mc_params(**params)
        sigma = np.ascontiguousarray(sigma, dtype=np.float64).reshape(-1, 4)
        n = sigma.shape[0]
        f = np.empty(n); g = np.empty(n); dg = np.empty((n, 4))
        self.lib.oracle_mc_surface(C.byref(prm), n, _dp(sigma), _dp(f), _dp(g), _dp(dg))
        return f, g, dg


_oracle = None


def load_oracle() -> OracleLib:
    global _oracle
    if _oracle is None:
        build_oracle()
        _oracle = OracleLib()
    return _oracle
