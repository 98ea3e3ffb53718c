#!/usr/bin/env python3
"""A load-stepping Newton-Krylov solve of a von Mises plasticity problem in which quadrature data never leaves the GPU.

The reference solves the same kind of problem (doc/demo/demo_plasticity_von_mises.py) with DOLFINx assembly + PETSc
SNES; per Newton iteration it evaluates the operands, calls the external operator, copies the tangent into a coefficient
and assembles a sparse Jacobian from it. Here, per Newton iteration:

    sigma, dp = von Mises(eps(Du), sigma_n, p)              dxo_von_mises_residual (strain + return map, then the internal force of
    R = sum w|J| B^T sigma  on the free dofs                                        the returned stress; no external load here)
    solve K d = -R with Jacobi-preconditioned CG, K v by    dxo_tangent_apply_vm, diag(K) by dxo_tangent_diagonal_vm   (K is never formed,
                                                            and neither is C_tang: both act from the returned (sigma, dp))
    Du += d
and at the end of a load step  p += dp, sigma_n <- sigma    dxo_vm_commit_state.
Only dof vectors (and a few scalars of the CG) are touched outside the kernels; they are torch CUDA tensors.

Problem: unit square (plane strain, P2 triangles), bottom edge clamped, top edge pulled upwards in steps.
Needs an MI355X.    python3 examples/device_newton_krylov.py [cells_per_side] [graph]
"""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dolfinx_external_operator_amd import MEM_DEVICE, Context, DeviceMesh, VmParams  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402


def main(n_side: int = 64, steps=(0.0012, 0.0024, 0.0036, 0.0048), verbose: bool = True, tangent_array: bool = False,
         graph: bool = False) -> dict:
    """tangent_array = False (default): the tangent block C_tang never exists — the fused operator writes (sigma, dp) only and the
    Krylov matvec / the Jacobi diagonal form the consistent tangent's action from them (dxo_tangent_apply_vm, dxo_tangent_diagonal_vm:
    56 instead of 288 bytes per point at d = 6). True: the operator writes C_tang as the reference's callback does and the matvec
    reads it (dxo_tangent_apply). graph = True: the CG iteration runs as a replayed HIP graph."""
    dev = torch.device("cuda:0")
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_option("consumer_overwrite", 1)      # residual / matvec / diagonal calls SET their output vector (no memset before each)
    mesh = structured_mesh("triangle", (n_side, n_side), degree=2)
    dm = DeviceMesh.from_synthetic(mesh, ctx=ctx)
    G, d = 2, 4
    nn, npts = mesh.node_x.shape[0], mesh.num_cells * mesh.nq
    E, nu, sigma_0 = 70e3, 0.3, 250.0
    prm = VmParams(E, nu, sigma_0, E * (E / 100) / (E - E / 100))          # demo constants, :185-188
    x = mesh.node_x
    bottom, top = x[:, 1] < 1e-12, x[:, 1] > 1 - 1e-12
    fixed = np.zeros((nn, G), dtype=bool)
    fixed[bottom] = True
    fixed[top] = True                                                        # top: prescribed (u_x = 0, u_y = load)
    free = torch.from_numpy(~fixed.reshape(-1)).to(dev)

    f64 = dict(dtype=torch.float64, device=dev)
    u, u_n = torch.zeros(nn * G, **f64), torch.zeros(nn * G, **f64)         # total displacement, last converged one
    sigma_n, p = torch.zeros(npts * d, **f64), torch.zeros(npts, **f64)     # state at the last converged step
    sigma, dp = torch.zeros(npts * d, **f64), torch.zeros(npts, **f64)
    C_tang = torch.zeros(npts * d * d, **f64) if tangent_array else None
    R, Kv = torch.zeros(nn * G, **f64), torch.zeros(nn * G, **f64)

    def residual(Du):
        """(sigma, dp[, C_tang]) of the increment and the internal force of the returned stress on the free dofs"""
        if tangent_array:
            dm.von_mises(prm, Du.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), C_tang.data_ptr(), sigma.data_ptr(), dp.data_ptr(), mem=MEM_DEVICE)
            dm.adjoint("eps", G, sigma.data_ptr(), R.data_ptr())
        else:
            dm.von_mises_residual(prm, Du.data_ptr(), sigma_n.data_ptr(), p.data_ptr(), sigma.data_ptr(), dp.data_ptr(), R.data_ptr())
        return torch.where(free, R, torch.zeros_like(R))

    def K_times(v):
        if tangent_array:
            dm.tangent_apply(C_tang.data_ptr(), v.data_ptr(), Kv.data_ptr())
        else:
            dm.tangent_apply_vm(prm, sigma.data_ptr(), dp.data_ptr(), v.data_ptr(), Kv.data_ptr())
        return torch.where(free, Kv, torch.zeros_like(Kv))

    diag = torch.zeros(nn * G, **f64)
    # persistent CG vectors: every scalar of an iteration stays on the device, so one iteration is a fixed sequence of launches —
    # with graph=True it is captured ONCE in a HIP graph (torch.cuda.graph; the library's device entry points are capture-safe after
    # their first call) and replayed: on this mesh size a CG iteration is launch-bound (about 70 us eager, 42 us replayed)
    xk, r, z, pk, Ap, minv = (torch.zeros(nn * G, **f64) for _ in range(6))
    rz = torch.zeros(1, **f64)
    captured = {}

    def cg_iteration():
        # The convergence test runs only every `check_every` iterations, so up to check_every - 1 iterations run on a system that
        # has already converged: there r . z and p . K p reach 0 and the two quotients would be 0 / 0. Both are guarded ON THE
        # DEVICE (a zero step / no new direction instead of NaN in xk and u); the captured graph replays the same guarded body.
        Ap.copy_(K_times(pk))
        pAp = torch.dot(pk, Ap)
        alpha = torch.where(pAp > 0, rz / pAp, torch.zeros_like(rz))
        xk.add_(alpha * pk)
        r.sub_(alpha * Ap)
        torch.mul(minv, r, out=z)
        rz_new = torch.dot(r, z).reshape(1)
        beta = torch.where(rz > 0, rz_new / rz, torch.zeros_like(rz))
        pk.mul_(beta).add_(z)
        rz.copy_(rz_new)

    def capture():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        saved = [t.clone() for t in (xk, r, z, pk, rz)]
        with torch.cuda.stream(side):                                   # warm-up on a side stream, as torch.cuda.graph asks for
            ctx.set_stream(side.cuda_stream)
            cg_iteration()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)     # the capture stream
            cg_iteration()
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for t, s0 in zip((xk, r, z, pk, rz), saved):
            t.copy_(s0)
        captured["graph"] = g

    def cg(b, tol=1e-10, maxit=4000, check_every=8):
        """Jacobi-preconditioned conjugate gradients; diag(K) comes from dxo_tangent_diagonal*, also matrix-free. The convergence test
        (the only host synchronisation) runs every `check_every` iterations."""
        if tangent_array:
            dm.tangent_diagonal(C_tang.data_ptr(), diag.data_ptr())
        else:
            dm.tangent_diagonal_vm(prm, sigma.data_ptr(), dp.data_ptr(), diag.data_ptr())
        minv.copy_(torch.where(free, 1.0 / diag, torch.zeros_like(diag)))
        xk.zero_()
        r.copy_(b)
        torch.mul(minv, r, out=z)
        pk.copy_(z)
        rz.copy_(torch.dot(r, z).reshape(1))
        b2 = float(torch.dot(b, b))
        if graph and "graph" not in captured:
            capture()
        its = 0
        while its < maxit and float(torch.dot(r, r)) > tol * tol * b2:
            for _ in range(check_every):
                if graph:
                    captured["graph"].replay()
                else:
                    cg_iteration()
            its += check_every
        return xk.clone(), its

    # predictor of a load step: homogeneous stretch u_y = load * y (satisfies both Dirichlet edges). It also keeps every
    # point away from the reference kernel's 0/0 at zero deviatoric stress (:318-319), which the demo avoids by
    # starting from Du = machine epsilon (:548-555).
    stretch = torch.zeros(nn * G, **f64)
    stretch[1::G] = torch.from_numpy(x[:, 1].copy()).to(dev)
    report = {"points": npts, "dofs": nn * G, "steps": []}
    prev = 0.0
    for load in steps:
        u += (load - prev) * stretch
        prev = load
        history, t0, cg_its = [], time.perf_counter(), 0
        for it in range(25):
            res = residual(u - u_n)
            rn = float(torch.linalg.norm(res))
            history.append(rn)
            if rn <= 1e-9 * max(history[0], 1e-30) or rn < 1e-9:
                break
            du, k = cg(-res)
            cg_its += k
            u += du
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ctx.vm_commit_state(d, npts, p.data_ptr(), dp.data_ptr(), sigma_n.data_ptr(), sigma.data_ptr())   # :564-565
        u_n.copy_(u)
        plastic = float((p > 0).double().mean())
        report["steps"].append({"load": load, "newton_residuals": history, "cg_iterations": cg_its, "seconds": dt,
                                "plastic_fraction": plastic, "max_p": float(p.max())})
        if verbose:
            print(f"load {load:.4f}: {len(history) - 1} Newton its, residuals " + " ".join(f"{r:.2e}" for r in history)
                  + f", {cg_its} CG its, {dt * 1e3:.0f} ms, plastic {plastic:.2f}")
    dm.close()
    ctx.close()
    return report


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 64, graph="graph" in sys.argv[2:])
