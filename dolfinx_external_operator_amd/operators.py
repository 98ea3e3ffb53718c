"""Drop-in `external_function` factories backed by the HIP kernels in libdxo_hip.so.

The reference's contract (src/dolfinx_external_operator/external_operator.py:432):

    values = external_operator.external_function(external_operator.derivatives)(*operand_arrays)

`external_function` maps the derivative multi-index to a callable; the callable receives one ndarray
per operand, shaped (num_cells, nq, *ufl_shape), and returns a flat array (or a tuple whose first
entry is assigned to the coefficient, :435-438). The factories below return objects that honour exactly
that contract, so a user of the reference only swaps

    sigma.external_function = sigma_external                 # demo_plasticity_von_mises.py:371
for
    sigma.external_function = make_von_mises(sigma_n, p)     # same call sites, same tuple order

State variables are NOT operands in the reference: `C_tang_impl` reads `sigma_n.x.array` / `p.x.array`
through its closure at every call (demo_plasticity_von_mises.py:347-348) because the load-stepping loop
mutates them (:564-565). The factories therefore take the *holders* (a `fem.Function`, an ndarray, or a
zero-argument callable) and re-read them at every evaluation.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from .operand_eval import LazyOperand
from ._lib import MEM_DEVICE, MEM_HOST, Context, IsiharaParams, McParams, VmParams, default_context


def _state_array(holder):
    """Current value of a closure-captured state variable (fem.Function | ndarray | callable | tensor)."""
    if callable(holder) and not hasattr(holder, "x") and not hasattr(holder, "data_ptr"):
        holder = holder()
    x = getattr(holder, "x", None)
    if x is not None and hasattr(x, "array"):
        return x.array  # dolfinx.fem.Function
    return holder


def _is_device_tensor(a) -> bool:
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _as_f64_host(a, what: str) -> np.ndarray:
    arr = np.asarray(a)
    if arr.dtype != np.float64:
        raise TypeError(f"{what}: the HIP kernels are fp64 (reference default PETSc.ScalarType); got {arr.dtype}")
    return np.ascontiguousarray(arr)


class _Outputs:
    """Output buffers for host calls.

    reuse=True (default of the factories): the returned flat arrays are views of pinned (hipHostMalloc) buffers
    owned by the operator and are OVERWRITTEN BY ITS NEXT CALL. That matches how the reference consumes them —
    `evaluate_external_operators` copies element 0 into the coefficient at once (external_operator.py:441) and
    the demos copy the extras right after the call (demo_plasticity_von_mises.py:451-456) — and it is what makes
    the boundary PCIe-bound instead of page-fault-bound: 7.5 ms vs 38 ms per call at 10^6 points (d = 6), because
    a fresh 344 MB ndarray is first touched inside the D2H copy. reuse=False returns fresh pageable arrays."""

    def __init__(self, ctx: Context, reuse: bool):
        self.ctx = ctx
        self.reuse = reuse
        self._cache: dict[tuple[str, int], np.ndarray] = {}

    def get(self, key: str, size: int) -> np.ndarray:
        if not self.reuse:
            return np.empty(size, dtype=np.float64)
        buf = self._cache.get((key, size))
        if buf is None:
            for k in [k for k in self._cache if k[0] == key]:   # the batch size changed: release the old buffer
                self.ctx.pinned_free(self._cache.pop(k))
            buf = self.ctx.pinned_empty(size)
            self._cache[(key, size)] = buf
        return buf


def make_von_mises(sigma_n, p, *, E: float = 70e3, nu: float = 0.3, sigma_0: float = 250.0,
                   H: float | None = None, ctx: Context | None = None, device: int = 0,
                   reuse_outputs: bool = True) -> Callable:
    """`sigma_external` of the von Mises demo (demo_plasticity_von_mises.py:364-368) on the GPU.

    Returns `external_function` with `external_function((1,))(deps) -> (C_tang, sigma, dp)`, flat arrays
    in the reference's order (:352). Any other multi-index raises NotImplementedError as in :367-368.
    `deps` has shape (num_cells, nq, d), d = 4 (reference) or 6 (3-D Mandel). Default constants: :185-188.
    Host ndarrays go through the chunked H2D/kernel/D2H pipeline; torch CUDA tensors stay on the device
    (outputs are then CUDA tensors on the same device, launched on torch's current stream).
    With reuse_outputs=True (default) the returned arrays live in pinned buffers that the NEXT call of this
    callable overwrites (see _Outputs); pass reuse_outputs=False for fresh arrays at ~5x the call time.
    """
    if H is None:
        E_tangent = E / 100.0                      # :186
        H = E * E_tangent / (E - E_tangent)        # :187
    prm = VmParams(float(E), float(nu), float(sigma_0), float(H))
    holder = {"ctx": ctx, "out": None}

    def _ctx() -> Context:
        if holder["ctx"] is None:
            holder["ctx"] = default_context(device)
        if holder["out"] is None:
            holder["out"] = _Outputs(holder["ctx"], reuse_outputs)
        return holder["ctx"]

    def C_tang_impl(deps):
        c = _ctx()
        if _is_device_tensor(deps):
            return _von_mises_device(c, prm, deps, _state_array(sigma_n), _state_array(p))
        if isinstance(deps, LazyOperand) and deps.kind == "eps" and deps.mesh.ctx is c:
            # operand still unevaluated: strain + return map + tangent in ONE launch (dxo_von_mises_field)
            n, d = deps.shape[0] * deps.shape[1], deps.shape[2]
            sigma_n_ = _as_f64_host(_state_array(sigma_n), "sigma_n").reshape(-1)
            p_ = _as_f64_host(_state_array(p), "p").reshape(-1)
            if sigma_n_.size != n * d or p_.size != n:
                raise ValueError(f"state size mismatch: sigma_n {sigma_n_.size} (want {n * d}), p {p_.size} (want {n})")
            out = holder["out"]
            C_tang_, sigma_, dp_ = out.get("C_tang", n * d * d), out.get("sigma", n * d), out.get("dp", n)
            deps.mesh.von_mises(prm, deps.u, sigma_n_, p_, C_tang_, sigma_, dp_)
            return C_tang_.reshape(-1), sigma_.reshape(-1), dp_.reshape(-1)
        deps = np.asarray(deps)
        num_cells, num_quadrature_points, d = deps.shape      # :344
        if d not in (4, 6):
            raise ValueError(f"von Mises kernel supports Mandel vectors of length 4 or 6, got {d}")
        n = num_cells * num_quadrature_points
        deps_ = _as_f64_host(deps, "deps").reshape(n, d)
        sigma_n_ = _as_f64_host(_state_array(sigma_n), "sigma_n").reshape(-1)
        p_ = _as_f64_host(_state_array(p), "p").reshape(-1)
        if sigma_n_.size != n * d or p_.size != n:
            # the reference's reshape at :347-348 raises ValueError on a size mismatch
            raise ValueError(f"state size mismatch: sigma_n {sigma_n_.size} (want {n * d}), p {p_.size} (want {n})")
        out = holder["out"]
        C_tang_ = out.get("C_tang", n * d * d)
        sigma_ = out.get("sigma", n * d)
        dp_ = out.get("dp", n)
        c.von_mises(prm, d, n, MEM_HOST, deps_, sigma_n_, p_, C_tang_, sigma_, dp_)
        return C_tang_.reshape(-1), sigma_.reshape(-1), dp_.reshape(-1)   # :352

    def sigma_external(derivatives):
        if derivatives == (1,):
            return C_tang_impl
        raise NotImplementedError(f"No external function is defined for the requested derivative {derivatives}.")

    sigma_external.params = prm
    sigma_external.context = _ctx
    return sigma_external


def _von_mises_device(c: Context, prm: VmParams, deps, sigma_n, p):
    import torch

    if deps.dtype != torch.float64:
        raise TypeError(f"deps: the HIP kernels are fp64, got {deps.dtype}")
    num_cells, nq, d = deps.shape
    n = num_cells * nq
    deps = deps.contiguous()
    sigma_n = sigma_n.contiguous()
    p = p.contiguous()
    if sigma_n.numel() != n * d or p.numel() != n:
        raise ValueError("state size mismatch")
    C_tang = torch.empty(n * d * d, dtype=torch.float64, device=deps.device)
    sigma = torch.empty(n * d, dtype=torch.float64, device=deps.device)
    dp = torch.empty(n, dtype=torch.float64, device=deps.device)
    c.set_stream(torch.cuda.current_stream(deps.device).cuda_stream)
    c.von_mises(prm, d, n, MEM_DEVICE, deps.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C_tang.data_ptr(),
                sigma.data_ptr(), dp.data_ptr())
    return C_tang, sigma, dp


def von_mises_commit_state(p, dp, sigma_n, sigma, *, ctx: Context | None = None, device: int = 0) -> None:
    """End-of-load-step history update for DEVICE-resident state (demo_plasticity_von_mises.py:564-565):
    `p += dp; sigma_n[:] = sigma` in one fused launch. Arguments are torch CUDA tensors (fp64, contiguous);
    p/dp hold n values, sigma_n/sigma n*d. Host-resident state keeps the reference's two NumPy statements."""
    import torch

    for name, t in (("p", p), ("dp", dp), ("sigma_n", sigma_n), ("sigma", sigma)):
        if not _is_device_tensor(t) or t.dtype != torch.float64 or not t.is_contiguous():
            raise TypeError(f"{name}: expected a contiguous fp64 CUDA tensor")
    n = p.numel()
    if dp.numel() != n or n == 0 and sigma.numel() != 0 or sigma_n.numel() != sigma.numel() or (n and sigma.numel() % n):
        raise ValueError("state size mismatch")
    d = sigma.numel() // n if n else 4
    c = ctx if ctx is not None else default_context(device)
    c.set_stream(torch.cuda.current_stream(p.device).cuda_stream)
    c.vm_commit_state(d, n, p.data_ptr(), dp.data_ptr(), sigma_n.data_ptr(), sigma.data_ptr())


def make_heat(*, A: float = 1.0, B: float = 1.0, ctx: Context | None = None, device: int = 0,
              fuse_by_identity: bool = False) -> Callable:
    """`q_external` of the nonlinear-heat demo (demo_nonlinear_heat_equation_part2.py:276-284) on the GPU.

    external_function((0, 0))(T, sigma) -> q        flat (N*gdim,)        (:219-230)
    external_function((1, 0))(T, sigma) -> dq/dT    flat (N*gdim,)        (:243-247)
    external_function((0, 1))(T, sigma) -> dq/dsig  flat (N*gdim*gdim,)   (:259-261)
    T: (num_cells, nq); sigma: (num_cells, nq*gdim) or (num_cells, nq, gdim) (:222-225); gdim is
    sigma.size / T.size.

    By default every call launches the kernel with only the requested output (the other output pointers
    are NULL). With fuse_by_identity=True the first call for a given pair of operand OBJECTS computes all
    three outputs in one launch and the next two calls (part2.py:307-309 evaluates F- and J-operators
    from the same `evaluated_operands` dict) are served from that result. Only enable it if the operand
    arrays are not modified in place between those calls (evaluate_operands returns fresh arrays).
    """
    holder = {"ctx": ctx, "keep": None, "val": None}

    def _eval(T, sigma, which: int):
        if holder["ctx"] is None:
            holder["ctx"] = default_context(device)
        c = holder["ctx"]
        if fuse_by_identity and holder["keep"] is not None and holder["keep"][0] is T and holder["keep"][1] is sigma:
            return holder["val"][which]
        if (isinstance(T, LazyOperand) and isinstance(sigma, LazyOperand) and T.kind == "value" and sigma.kind == "grad"
                and T.bs == 1 and sigma.bs == 1 and T.mesh is sigma.mesh and T.mesh.ctx is c and np.array_equal(T.u, sigma.u)):
            # both operands are still unevaluated views of the SAME scalar field: T, grad T and the requested output
            # in one launch (dxo_heat_field)
            n, gdim = T.shape[0] * T.shape[1], T.mesh.gdim
            sizes = (n * gdim, n * gdim, n * gdim * gdim)
            outs = [np.empty(sz) if k == which else None for k, sz in enumerate(sizes)]
            T.mesh.heat(A, B, T.u, outs[0], outs[1], outs[2])
            return outs[which]
        T_ = _as_f64_host(T, "T").reshape(-1)
        n = T_.size
        sig_ = _as_f64_host(sigma, "sigma").reshape(-1)
        if n == 0:
            gdim = sigma.shape[-1] if np.ndim(sigma) == 3 and sigma.shape[-1] in (1, 2, 3) else 2
        else:
            if sig_.size % n:
                raise ValueError(f"sigma size {sig_.size} is not a multiple of the number of points {n}")
            gdim = sig_.size // n
        sizes = (n * gdim, n * gdim, n * gdim * gdim)
        outs = [np.empty(sz) if (fuse_by_identity or k == which) else None for k, sz in enumerate(sizes)]
        c.heat(A, B, gdim, n, MEM_HOST, T_, sig_, outs[0], outs[1], outs[2])
        if fuse_by_identity:
            holder["keep"] = (T, sigma)
            holder["val"] = outs
        return outs[which]

    def q_impl(T, sigma):
        return _eval(T, sigma, 0)

    def dqdT_impl(T, sigma):
        return _eval(T, sigma, 1)

    def dqdsigma_impl(T, sigma):
        return _eval(T, sigma, 2)

    def q_external(derivatives):
        if derivatives == (0, 0):
            return q_impl
        elif derivatives == (1, 0):
            return dqdT_impl
        elif derivatives == (0, 1):
            return dqdsigma_impl
        raise NotImplementedError(f"No external function is defined for the requested derivative {derivatives}.")

    return q_external


def make_mohr_coulomb(sigma_n, *, E: float = 6778.0, nu: float = 0.25, c: float = 3.45,
                      phi: float = 30 * np.pi / 180, psi: float = 30 * np.pi / 180, theta_T: float = 26 * np.pi / 180,
                      a: float | None = None, tol: float = 1e-8, Nitermax: int = 200, diagnostics: bool = True,
                      on_summary: Callable | None = None, ctx: Context | None = None, device: int = 0,
                      reuse_outputs: bool = True) -> Callable:
    """`sigma_external` of the Mohr-Coulomb demo (demo_plasticity_mohr_coulomb.py:604-608) on the GPU.

    `external_function((1,))(deps) -> (C_tang, sigma)`, flat arrays, the reference's order (:593); any other
    multi-index raises NotImplementedError (:607-608). `deps` is reshaped to (-1, 4) (:578); `sigma_n` is the
    closure state (a fem.Function, ndarray or callable), re-read at every call (:579, :728).
    Defaults are the demo's constants (:110-116) and Newton controls (:469).

    The reference prints an "Inner Newton summary" at every call (:584-591). Here the four per-point aux
    arrays (niter, yielding, norm_res, dlambda; :533) are kept on `external_function.last_state` when
    `diagnostics=True`, and `on_summary(dict)` — if given — receives the same numbers the reference prints:
    unique iteration counts, their multiplicities, max f, max residual.
    """
    if a is None:
        a = 0.26 * c / np.tan(phi)   # :116
    prm = McParams(float(E), float(nu), float(c), float(phi), float(psi), float(theta_T), float(a), float(tol),
                   int(Nitermax), 0)
    holder = {"ctx": ctx, "out": None}

    def C_tang_impl(deps):
        if holder["ctx"] is None:
            holder["ctx"] = default_context(device)
        cx = holder["ctx"]
        if holder["out"] is None:
            holder["out"] = _Outputs(cx, reuse_outputs)
        deps_ = _as_f64_host(deps, "deps").reshape((-1, 4))            # :578
        sigma_n_ = _as_f64_host(_state_array(sigma_n), "sigma_n").reshape((-1, 4))   # :579
        n = deps_.shape[0]
        if sigma_n_.shape[0] != n:
            raise ValueError(f"state size mismatch: sigma_n has {sigma_n_.shape[0]} points, deps {n}")
        C_tang = holder["out"].get("C_tang", n * 16)
        sigma = holder["out"].get("sigma", n * 4)
        if diagnostics or on_summary is not None:
            niter = np.empty(n, dtype=np.int32)
            yielding, norm_res, dlambda = np.empty(n), np.empty(n), np.empty(n)
        else:
            niter = yielding = norm_res = dlambda = None
        cx.mohr_coulomb(prm, n, MEM_HOST, deps_, sigma_n_, C_tang, sigma, niter, yielding, norm_res, dlambda)
        sigma_external.last_state = (niter, yielding, norm_res, dlambda)
        if on_summary is not None and n > 0:
            unique_iters, counts = np.unique(niter, return_counts=True)   # :584
            on_summary({"unique_iters": unique_iters, "counts": counts, "max_yielding": float(np.max(yielding)),
                        "max_norm_res": float(np.nanmax(norm_res)) if np.isfinite(norm_res).any() else float("nan")})
        return C_tang.reshape(-1), sigma.reshape(-1)                    # :593

    def sigma_external(derivatives):
        if derivatives == (1,):
            return C_tang_impl
        raise NotImplementedError(f"No external function is defined for the requested derivative {derivatives}.")

    sigma_external.params = prm
    sigma_external.last_state = None
    return sigma_external


def make_icnn(state_dict, *, precision: str = "fp32", ctx: Context | None = None, device: int = 0,
              reuse_outputs: bool = True) -> Callable:
    """`P_external` of the hyperelasticity demo (demo_hyperelasticity.py:459-466) on the GPU.

    `external_function((1,))(Fvals) -> (dP, P)`, flat arrays in the reference's order (:456); other
    multi-indices raise NotImplementedError. `Fvals` is (num_cells, nq, 2, 2) (or anything reshaping to
    (-1, 4), :452). `state_dict` is the model's `state_dict()` (torch tensors or arrays; keys as torch names
    them or with '__' for '.'), e.g. `torch.load("Isihara_noise=high.pth")` (:314).
    precision "fp32" evaluates the network in fp32 like the reference (:286); "fp64" is the tolerance-study
    variant (BASELINE config 5). The stress correction H (:362-381) is computed once at creation.
    """
    prec = {"fp32": 0, "fp64": 1}[precision]
    holder = {"ctx": ctx, "model": None, "out": None}

    def _model():
        if holder["ctx"] is None:
            holder["ctx"] = default_context(device)
        if holder["model"] is None:
            holder["model"] = holder["ctx"].icnn_create(state_dict)
            holder["out"] = _Outputs(holder["ctx"], reuse_outputs)
        return holder["ctx"], holder["model"]

    def dP_dF_impl(Fvals):
        cx, model = _model()
        F = _as_f64_host(Fvals, "Fvals").reshape(-1, 4)      # :452
        n = F.shape[0]
        dP, P = holder["out"].get("dP", n * 16), holder["out"].get("P", n * 4)
        cx.icnn_eval(model, prec, n, MEM_HOST, F, dP, P)
        return dP.reshape(-1), P.reshape(-1)                 # :456

    def P_external(derivatives):
        if derivatives == (1,):
            return dP_dF_impl
        raise NotImplementedError(f"No external function is defined for the requested derivative {derivatives}.")

    P_external.correction = lambda: _model()[0].icnn_correction(_model()[1])
    return P_external


__all__ = ["make_von_mises", "make_heat", "make_mohr_coulomb", "make_icnn"]


def make_isihara(*, c1: float = 0.5, c2: float = 1.0, c3: float = 1.0, c4: float = 1.5, ctx: Context | None = None,
                 device: int = 0, reuse_outputs: bool = True) -> Callable:
    """The analytic Isihara model behind the same `P_external` contract as `make_icnn`.

    The reference states it in UFL only (demo_hyperelasticity.py:686-703, `P = ufl.diff(W_Isihara, F_)`) and
    uses it as the ground truth the network is compared with (:806-817). As an external operator it takes the
    operand `F = I + grad u` exactly like the network: `external_function((1,))(Fvals) -> (dP, P)`.
    W = c1 (I1bar-3) + c2 (I2bar-3) + c3 (I1bar-3)^2 + c4 (J-1)^2; defaults are the reference's (:700)."""
    prm = IsiharaParams(float(c1), float(c2), float(c3), float(c4))
    holder = {"ctx": ctx, "out": None}

    def dP_dF_impl(Fvals):
        if holder["ctx"] is None:
            holder["ctx"] = default_context(device)
        if holder["out"] is None:
            holder["out"] = _Outputs(holder["ctx"], reuse_outputs)
        F = _as_f64_host(Fvals, "Fvals").reshape(-1, 4)
        n = F.shape[0]
        dP, P = holder["out"].get("dP", n * 16), holder["out"].get("P", n * 4)
        holder["ctx"].isihara(prm, n, MEM_HOST, F, dP, P)
        return dP.reshape(-1), P.reshape(-1)

    def P_external(derivatives):
        if derivatives == (1,):
            return dP_dF_impl
        raise NotImplementedError(f"No external function is defined for the requested derivative {derivatives}.")

    P_external.params = prm
    return P_external
