#!/usr/bin/env python3
"""Device operand evaluation timing (GPU box): strain at quadrature points from a P2/Q2 displacement field.
usage: python3 scripts/bench_operand.py [--cpu 1]
Each line is one JSON object. Algorithmic bytes per launch = field vector + coordinates (each node read once) +
both dofmaps + the output array."""
import argparse
import json
import pathlib
import statistics
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from dolfinx_external_operator_amd import Context, DeviceMesh  # noqa: E402
from tools.synthetic import structured_mesh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cpu", type=int, default=0, help="also time the NumPy oracle on the first 50 000 cells")
ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--case", type=int, default=-1, help="run only this case (index into CASES); -1 = all")
ap.add_argument("--operand-cell", type=int, default=-1, help="ctx option operand_cell: 1 lane = cell kernels, 0 wave-group kernels; -1 = both")
args = ap.parse_args()

dev = torch.device("cuda:0")
ctx = Context(0)
stream = torch.cuda.current_stream()
ctx.set_stream(stream.cuda_stream)

CASES = [("hexahedron", (108, 108, 108), 2, "eps", "Q2 hexahedra, 8 qp/cell, eps(u) Mandel d=6 (BASELINE config 2 layout, 1e7 points)"),
         ("hexahedron", (50, 50, 50), 2, "eps", "Q2 hexahedra, 50^3 cells = 1e6 points (BASELINE config 2)"),
         ("triangle", (1000, 1000), 2, "eps", "P2 triangles, 3 qp/cell, eps(u) Mandel d=4 (the reference demos' layout)"),
         ("triangle", (1000, 1000), 2, "F", "P2 triangles, F = I + grad u (hyperelasticity demo operand)")]
for cell, n, degree, kind, label in (CASES if args.case < 0 else [CASES[args.case]]):
  m = structured_mesh(cell, n, degree, distort=0.2, seed=0)
  for oc in ((1, 0) if args.operand_cell < 0 else (args.operand_cell,)):
    ctx.set_option("operand_cell", oc)
    label = label.split(" [")[0] + (" [lane = cell kernels]" if oc else " [wave-group kernels]")
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    bs = m.gdim
    D = dm.value_size(kind, bs)
    rng = np.random.Generator(np.random.PCG64(0))
    u_h = rng.normal(0.0, 1e-3, size=m.node_x.shape[0] * bs)
    u = torch.from_numpy(u_h).to(dev)
    out = torch.empty(m.num_cells * m.nq * D, dtype=torch.float64, device=dev)
    run = lambda: dm.evaluate_device(kind, bs, u.data_ptr(), m.num_cells, out.data_ptr())  # noqa: E731
    for _ in range(3):
        run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
    for a, b in ev:
        a.record(stream); run(); b.record(stream)
    torch.cuda.synchronize()
    ms = statistics.median(a.elapsed_time(b) for a, b in ev)
    npts = m.num_cells * m.nq
    bytes_alg = u_h.nbytes + m.x.nbytes + m.dofmap.nbytes + m.geom_dofmap.nbytes + npts * D * 8
    flop = npts * (2 * m.dofmap.shape[1] * bs * m.gdim + 2 * m.geom_dofmap.shape[1] * m.gdim ** 2 + 2 * bs * m.gdim ** 2 + 40)
    rec = {"case": label, "cells": m.num_cells, "points": npts, "value_size": D, "kernel_ms": ms, "qp_per_s": npts / ms * 1e3,
           "GBps_algorithmic": bytes_alg / ms / 1e6, "bytes_per_qp": bytes_alg / npts, "GFLOPs_fp64": flop / ms / 1e6}
    if args.cpu:
        from oracle.operand_oracle import eval_operand
        nc = min(50_000, m.num_cells)
        cells = np.arange(nc)
        t0 = time.perf_counter()
        ref = eval_operand({"eps": 2, "F": 3}[kind], bs, u_h, m.dofmap, m.geom_dofmap, m.x, m.phi, m.dphi, m.dpsi, cells)
        dt = time.perf_counter() - t0
        got = out[: nc * m.nq * D].cpu().numpy().reshape(ref.shape)
        rec["cpu_baseline"] = {"value": nc * m.nq / dt, "unit": "qp/s", "kind": "port", "cores": "numpy/BLAS default",
                               "sample": f"first {nc} cells, oracle/operand_oracle.py (einsum push-forward)"}
        rec["max_abs_diff_vs_oracle_on_sample"] = float(np.abs(got - ref).max())
    print(json.dumps(rec), flush=True)
    if kind == "eps":
        # the demo's pair evaluate_operands + evaluate_external_operators (demo_plasticity_von_mises.py:445-456):
        # two launches (operand, then dxo_von_mises reading it back) against the fused dxo_von_mises_field
        from dolfinx_external_operator_amd import MEM_DEVICE, VmParams
        E = 70e3
        prm = VmParams(E, 0.3, 250.0, E * (E / 100) / (E - E / 100))
        d = D
        g = torch.Generator(device=dev); g.manual_seed(1)
        sig = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64) * 100
        pp = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
        u.mul_(3.0)
        Ct = torch.empty(npts * d * d, dtype=torch.float64, device=dev)
        st = torch.empty(npts * d, dtype=torch.float64, device=dev)
        dpt = torch.empty(npts, dtype=torch.float64, device=dev)

        def two():
            dm.evaluate_device(kind, bs, u.data_ptr(), m.num_cells, out.data_ptr())
            ctx.von_mises(prm, d, npts, MEM_DEVICE, out.data_ptr(), sig.data_ptr(), pp.data_ptr(), Ct.data_ptr(), st.data_ptr(), dpt.data_ptr())

        def fused():
            dm.von_mises(prm, u.data_ptr(), sig.data_ptr(), pp.data_ptr(), Ct.data_ptr(), st.data_ptr(), dpt.data_ptr(), mem=MEM_DEVICE)

        res = {}
        for name, fn in (("two_launches", two), ("fused", fused)):
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
            for a, b in ev:
                a.record(stream); fn(); b.record(stream)
            torch.cuda.synchronize()
            res[name] = statistics.median(a.elapsed_time(b) for a, b in ev)
        out_bytes = npts * (d * d + d + 1) * 8
        in_bytes = npts * (d + 1) * 8 + u_h.nbytes + m.x.nbytes + m.dofmap.nbytes + m.geom_dofmap.nbytes
        print(json.dumps({"case": "operand + von Mises: " + label, "points": npts, "d": d, "plastic_fraction": float((dpt > 0).double().mean()),
                          "two_launches_ms": res["two_launches"], "fused_ms": res["fused"],
                          "fused_qp_per_s": npts / res["fused"] * 1e3, "fused_bytes_per_qp": (in_bytes + out_bytes) / npts,
                          "fused_GBps_algorithmic": (in_bytes + out_bytes) / res["fused"] / 1e6}), flush=True)
        # consumer side: internal force f = sum w|detJ| B^T sigma and the matrix-free tangent action K v
        fv = torch.zeros(m.node_x.shape[0] * bs, dtype=torch.float64, device=dev)
        vv = torch.randn(m.node_x.shape[0] * bs, dtype=torch.float64, device=dev)
        cons = {}
        for name, fn in (("internal_force", lambda: dm.adjoint("eps", bs, st.data_ptr(), fv.data_ptr())),
                         ("tangent_apply", lambda: dm.tangent_apply(Ct.data_ptr(), vv.data_ptr(), fv.data_ptr()))):
            for _ in range(3):
                fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
            for a, b in ev:
                a.record(stream); fn(); b.record(stream)
            torch.cuda.synchronize()
            cons[name] = statistics.median(a.elapsed_time(b) for a, b in ev)
        geo = m.x.nbytes + m.dofmap.nbytes + m.geom_dofmap.nbytes
        print(json.dumps({"case": "consumer side: " + label, "points": npts, "d": d,
                          "internal_force_ms": cons["internal_force"], "internal_force_qp_per_s": npts / cons["internal_force"] * 1e3,
                          "internal_force_GBps_algorithmic": (npts * d * 8 + geo + 2 * u_h.nbytes) / cons["internal_force"] / 1e6,
                          "tangent_apply_ms": cons["tangent_apply"], "tangent_apply_qp_per_s": npts / cons["tangent_apply"] * 1e3,
                          "tangent_apply_GBps_algorithmic": (npts * d * d * 8 + geo + 3 * u_h.nbytes) / cons["tangent_apply"] / 1e6}), flush=True)
        del sig, pp, Ct, st, dpt, fv, vv
    dm.close()
    del u, out
ctx.close()
