"""bench.py's `secondary.device_loop_*` legs: one Newton iteration of the reference's hot loop with quadrature data resident
in HBM, and the dofmap assigner.

The reference's loop is  evaluate_operands -> evaluate_external_operators -> assemble residual / Jacobian
(doc/demo/demo_plasticity_von_mises.py:445-456, src/dolfinx_external_operator/petsc/petsc.py:55-68). On the device it is
four library calls on device pointers (no PCIe traffic, only dof vectors leave the kernels):

    dxo_von_mises_field_state   eps(Du) + radial return + consistent tangent, history variables in a dxo_vm_state
    dxo_operand_adjoint         internal force  R = sum_q w|detJ| B^T sigma         (the residual form, :378-385)
    dxo_tangent_apply           K v = sum_q w|detJ| B^T C_tang B v, K never formed  (one Krylov matvec; the Jacobian form :386-391)
    dxo_vm_state_commit         p += dp, sigma_n <- sigma                           (:564-565; once per load step)

Every call is timed with HIP events on the launch stream and carries its own `roofline` (HBM; algorithmic bytes in
DESIGN.md 9.1, restated in `bytes` below). `iteration_ms` = field + internal force + one matvec. The cpu_baseline of the
iteration is the same three steps with the checkers (NumPy operand / adjoint oracles + the C return map) on a bounded
sample of the same mesh. Only that part touches oracle/.
"""
from __future__ import annotations

import time

import numpy as np

from tools.bench_secondary import ROOT, _avail, _hbm, _time  # noqa: F401


def _geo_bytes(m):
    return m.x.nbytes + m.geom_dofmap.nbytes


def loop_bytes(m, bs, d):
    """Algorithmic HBM bytes per call (every array once; `out` is written only: the legs run with option consumer_overwrite = 1)."""
    npts = m.num_cells * m.nq
    vec = m.node_x.shape[0] * bs * 8
    geo = _geo_bytes(m)
    return {
        "von_mises_field_state": vec + geo + m.dofmap.nbytes + npts * (d + 1) * 8 + npts * (d * d + d + 1) * 8,
        "internal_force": npts * d * 8 + geo + m.dofmap.nbytes + vec,
        "tangent_apply": npts * d * d * 8 + geo + m.dofmap.nbytes + vec + vec,
        "tangent_diagonal": npts * d * d * 8 + geo + m.dofmap.nbytes + vec,
        "state_commit": npts * (2 * d + 3) * 8,     # read sigma, dp, p; write sigma_n, p  (sigma_n is overwritten, not read)
    }


def field_dofs(m, rng):
    """Nodal values of a displacement increment whose strains are of the size SURVEY.md 8(d) prescribes (N(0, 3e-3) per component):
    independent N(0, s) values per dof with s = 1.2e-3 x the cell size (the gradient of nodal noise on a degree-2 element is about
    2.5 / h times its standard deviation), so that — with sigma_n ~ N(0, 100) — the batch is a mix of elastic and plastic points."""
    h = 1.0 / round(m.num_cells / (2 if m.cell == "triangle" else 1)) ** (1.0 / m.gdim)
    return rng.normal(0.0, 1.2e-3 * h, size=m.node_x.shape[0] * m.gdim)


def device_loop(torch, ctx, stream, prm, cell, n, cpu=True, launches=10):
    from dolfinx_external_operator_amd import MEM_DEVICE, DeviceMesh
    from tools.synthetic import structured_mesh_cached

    dev = torch.device("cuda", ctx.device)
    t0 = time.perf_counter()
    m = structured_mesh_cached(cell, n, 2, distort=0.2, seed=0)
    mesh_s = time.perf_counter() - t0
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    st = None
    ctx.set_option("consumer_overwrite", 1)     # the consumer-side calls SET their output vector: no memset in front of a matvec
    try:
        bs = m.gdim
        d = 4 if bs == 2 else 6
        npts, nn = m.num_cells * m.nq, m.node_x.shape[0]
        rng = np.random.Generator(np.random.PCG64(0))
        u_h = field_dofs(m, rng)
        u = torch.from_numpy(u_h).to(dev)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        sig0 = torch.randn(npts * d, generator=g, device=dev, dtype=torch.float64) * 100
        p0 = (torch.randn(npts, generator=g, device=dev, dtype=torch.float64) * 1e-3).abs()
        v = torch.randn(nn * bs, generator=g, device=dev, dtype=torch.float64)
        R = torch.zeros(nn * bs, dtype=torch.float64, device=dev)
        Kv = torch.zeros(nn * bs, dtype=torch.float64, device=dev)
        st = ctx.vm_state(d, npts)
        st.upload(sig0.data_ptr(), p0.data_ptr(), mem=MEM_DEVICE)
        ptr = st.pointers()
        by = loop_bytes(m, bs, d)
        # C_tang in an arena block chosen by timing the producing kernel on the candidates (as the headline's outputs are)
        (Ct,) = ctx.output_tensors_probed((npts * d * d,), lambda ptrs, shape: st.call_field(prm, dm._h, MEM_DEVICE, u.data_ptr(), ptrs[0]),
                                          bytes_per_launch=float(by["von_mises_field_state"]))

        def field():
            st.call_field(prm, dm._h, MEM_DEVICE, u.data_ptr(), Ct.data_ptr())

        def force():
            dm.adjoint("eps", bs, ptr["sigma"], R.data_ptr())

        def matvec():
            dm.tangent_apply(Ct.data_ptr(), v.data_ptr(), Kv.data_ptr())

        def diag():
            dm.tangent_diagonal(Ct.data_ptr(), Kv.data_ptr())

        def iteration():
            field()
            force()
            matvec()

        # the same iteration WITHOUT a tangent array: the fused operator writes (sigma, dp) only, K v and diag(K) come from them
        def field_state_only():
            st.call_field(prm, dm._h, MEM_DEVICE, u.data_ptr(), None)

        def matvec_vm():
            dm.tangent_apply_vm(prm, ptr["sigma"], ptr["dp"], v.data_ptr(), Kv.data_ptr())

        def diag_vm():
            dm.tangent_diagonal_vm(prm, ptr["sigma"], ptr["dp"], Kv.data_ptr())

        def iteration_vm():
            field_state_only()
            force()
            matvec_vm()

        field()
        torch.cuda.synchronize()
        plastic = float((torch.as_tensor(_view(torch, ptr["dp"], npts, dev)) > 0).double().mean())
        calls = {}
        for name, fn, kernels in (("von_mises_field_state", field, [f"vm_field<{bs}>"]),
                                  ("internal_force", force, ["operand_adjoint_c8 (hexahedra) | adjoint_cell_eps", "node_sum"]),
                                  ("tangent_apply", matvec, ["tangent_apply", "node_sum"]),
                                  ("tangent_diagonal", diag, ["tangent_diag", "node_sum"])):
            ms, _ = _time(torch, stream, fn, launches, warm=3)
            calls[name] = {"ms_per_call": ms, "qp_per_s": npts / ms * 1e3, "kernels": kernels,
                           "roofline": {**_hbm(by[name], ms), "algorithmic_bytes_per_call": by[name], "bytes_per_qp": by[name] / npts}}
        vec_b = nn * bs * 8
        by_vm = {"von_mises_field_state_no_tangent": by["von_mises_field_state"] - npts * d * d * 8,
                 "tangent_apply_vm": npts * (d + 1) * 8 + _geo_bytes(m) + m.dofmap.nbytes + 2 * vec_b,
                 "tangent_diagonal_vm": npts * (d + 1) * 8 + _geo_bytes(m) + m.dofmap.nbytes + vec_b}
        matvec()
        ref_Kv = Kv.clone()
        matvec_vm()
        vm_err = float((Kv - ref_Kv).abs().max() / ref_Kv.abs().max())
        calls_vm = {}
        for name, fn, kernels in (("von_mises_field_state_no_tangent", field_state_only, [f"vm_field<{bs}, ..., 1> (C_tang = NULL: its own instantiation)"]),
                                  ("tangent_apply_vm", matvec_vm, ["tangent_apply<..., VM>", "node_sum"]),
                                  ("tangent_diagonal_vm", diag_vm, ["tangent_diag<..., VM>", "node_sum"])):
            ms, _ = _time(torch, stream, fn, launches, warm=3)
            calls_vm[name] = {"ms_per_call": ms, "qp_per_s": npts / ms * 1e3, "kernels": kernels,
                              "roofline": {**_hbm(by_vm[name], ms), "algorithmic_bytes_per_call": by_vm[name], "bytes_per_qp": by_vm[name] / npts}}
        ms_it_vm, _ = _time(torch, stream, iteration_vm, launches, warm=2)
        field()        # leave the full outputs (with tangent) in place for the legs below
        # the load-step update: commit needs a fresh result each time, so the pair (field, commit) is timed and the field's time subtracted
        ms_pair, _ = _time(torch, stream, lambda: (field(), st.commit()), launches, warm=2)
        ms_c = max(ms_pair - calls["von_mises_field_state"]["ms_per_call"], 1e-6)
        calls["state_commit"] = {"ms_per_call": ms_c, "kernels": ["vm_commit"], "timed_as": "(field + commit) - field",
                                 "roofline": {**_hbm(by["state_commit"], ms_c), "algorithmic_bytes_per_call": by["state_commit"],
                                              "bytes_per_qp": by["state_commit"] / npts}}
        st.upload(sig0.data_ptr(), p0.data_ptr(), mem=MEM_DEVICE)
        ms_it, _ = _time(torch, stream, iteration, launches, warm=2)
        it_bytes = by["von_mises_field_state"] + by["internal_force"] + by["tangent_apply"]
        out = {"workload": f"one device-resident Newton iteration of the von Mises problem (petsc.py:55-68 / demo_plasticity_von_mises.py:445-456): "
                           f"fused operand + return map with resident state, internal force, one matrix-free tangent matvec; "
                           f"{cell} degree 2, {m.num_cells} cells x {m.nq} points = {npts} points, {nn * bs} dofs, Mandel d={d}, fp64",
               "points": npts, "dofs": nn * bs, "cells": m.num_cells, "value": npts / ms_it * 1e3, "unit": "qp/s per Newton iteration (1 matvec)",
               "iteration_ms": ms_it, "dtype": "f64", "plastic_fraction": plastic, "mesh_build_s": mesh_s, "calls": calls,
               "roofline": {**_hbm(it_bytes, ms_it), "algorithmic_bytes_per_iteration": it_bytes, "bytes_per_qp": it_bytes / npts,
                            "note": "sum of the three calls' algorithmic bytes over the iteration's time; each call has its own roofline under `calls`"},
               "pcie_bytes_per_iteration": 0, "consumer_overwrite": 1,
               "without_tangent_array": {
                   "meaning": "the same iteration with the von Mises tangent's action formed from the returned (sigma, dp) (dxo_tangent_apply_vm): "
                              "the fused operator runs with C_tang = NULL, a Krylov matvec reads 56 instead of 288 bytes per point",
                   "iteration_ms": ms_it_vm, "value": npts / ms_it_vm * 1e3, "calls": calls_vm,
                   "matvec_max_rel_diff_vs_tangent_array": vm_err}}
        _attach_fp64(out, "device_loop_q2hex" if cell == "hexahedron" else "device_loop_p2tri", bs)
        if cpu:
            out["cpu_baseline"] = _cpu_iteration(m, bs, d, u_h, sig0, p0, v, prm)
        return out
    finally:
        ctx.set_option("consumer_overwrite", 0)
        if st is not None:
            st.close()
        dm.close()


FP64_PEAK_TFLOPS = 78.6      # fp64 vector peak of the MI355X (MI355X_MICROARCH.md)


def _attach_fp64(out, leg, bs):
    """The element kernels of the consumer side are bound by fp64 arithmetic, not by HBM: next to every call's HBM roofline goes the fp64 work
    the kernel ISSUES per launch (counted on the kernel itself by scripts/profile_round.sh, profiles/consumer_flop.json: masked lanes included)
    over the call's measured time — a lower bound of the kernel's own rate, since the call also contains node_sum — and the kernel-only
    fraction of the profile run. Only when this mesh has the size the counts were taken on."""
    import json

    from tools.bench_secondary import _name_has

    f = ROOT / "profiles" / "consumer_flop.json"
    if not f.exists():
        return
    rec = json.loads(f.read_text())
    if rec.get("points", {}).get(leg) != out["points"]:
        return
    g = "3, 27, 8" if bs == 3 else "2, *"        # triangles: generic kernels <2, 0, 0, ..> or the matrix-pipe instantiation <2, 6, 3, ..>
    gd = "3, 27" if bs == 3 else "2, *"
    keys = {("calls", "von_mises_field_state"): f"vm_field<{bs},*, 0>", ("calls", "tangent_apply"): f"tangent_apply<{g}, false, *>",
            ("calls", "tangent_diagonal"): f"tangent_diag<{gd}, false, *>",
            ("calls", "internal_force"): "operand_adjoint_c8" if bs == 3 else "adjoint_cell_eps<2,",
            ("without_tangent_array", "calls", "von_mises_field_state_no_tangent"): f"vm_field<{bs},*, 1>",
            ("without_tangent_array", "calls", "tangent_apply_vm"): f"tangent_apply<{g}, true, *>",
            ("without_tangent_array", "calls", "tangent_diagonal_vm"): f"tangent_diag<{gd}, true, *>"}
    for path, key in keys.items():
        e = next((v for k, v in rec["fp64_issued"].items() if _name_has(key, k)), None)
        call = out
        for p_ in path:
            call = call.get(p_, {}) if isinstance(call, dict) else {}
        if not e or "ms_per_call" not in call:
            continue
        tf = e["fp64_flop_per_launch"] / call["ms_per_call"] / 1e9
        call["roofline"]["fp64_valu"] = {"bound": "fp64_valu", "flop_per_launch_issued": e["fp64_flop_per_launch"], "achieved_over_call": tf, "unit": "TFLOP/s",
                                         "peak": FP64_PEAK_TFLOPS, "frac_over_call": tf / FP64_PEAK_TFLOPS,
                                         "kernel_only_frac_in_profile_run": e.get("frac_of_78.6_TFLOP_per_s_fp64_vector_peak"),
                                         "source": "profiles/consumer_flop.json (" + rec.get("measured", "") + ")"}
        if e.get("fp64_mfma_flop_per_launch"):      # the scatter on the matrix pipe (cell8_mfma.h): beside, not part of, the vector-pipe figure
            call["roofline"]["fp64_valu"]["mfma_flop_per_launch_issued"] = e["fp64_mfma_flop_per_launch"]


def _view(torch, ptr, n, dev):
    """float64 tensor over n doubles of library-owned device memory (no copy)."""
    from dolfinx_external_operator_amd._lib import _CudaArrayView

    return torch.as_tensor(_CudaArrayView(None, int(ptr), int(n), "<f8"), device=dev)


def _cpu_iteration(m, bs, d, u_h, sig0, p0, v, prm, cells=500_000):
    """The same three steps in compiled, threaded C on the first `cells` cells of the same mesh: oracle/operand_oracle_c.c (strain at the
    points, internal force, tangent action; OpenMP over cells) and oracle/dxo_oracle.c (return map). The reference's own code for these
    steps is compiled too (DOLFINx / FFCx behind Expression.eval and assemble_vector, the Numba kernel). A few seconds of CPU work: the
    thread counts are scanned and the fastest is reported."""
    from oracle import load_oracle

    o = load_oracle()
    nc = min(cells, m.num_cells)
    nq = m.nq
    s_h = sig0[: nc * nq * d].cpu().numpy().reshape(nc * nq, d)
    p_h = p0[: nc * nq].cpu().numpy()
    v_h = v.cpu().numpy()
    nn = m.node_x.shape[0]
    n = nc * nq
    best = None
    scan = {}
    for nt in sorted({1, 8, min(32, _avail()), min(64, _avail())}):
        if nt == 1 and nc > 100_000:
            sub = 100_000                                  # one core: a tenth of the sample is enough for a rate
        else:
            sub = nc
        outb = tuple(np.full(shape, 0.5) for shape in ((sub * nq, d, d), (sub * nq, d), (sub * nq,)))      # written once: page faults stay out of the timing
        w = min(sub, 2000)                                  # warm-up of every step on a few cells (thread pool, instruction cache)
        dw = o.operand_eps(m, u_h, cells=w, nthreads=nt)
        Cw, sw, _ = o.von_mises(dw.reshape(-1, d), s_h[: w * nq], p_h[: w * nq], E=prm.E, nu=prm.nu, sigma_0=prm.sigma_0, H=prm.H, nthreads=nt)
        o.operand_eps_adjoint(m, sw.reshape(w, nq, d), nn, cells=w, nthreads=nt)
        o.tangent_apply(m, Cw, v_h, nn, cells=w, nthreads=nt)
        t0 = time.perf_counter()
        deps = o.operand_eps(m, u_h, cells=sub, nthreads=nt)
        t1 = time.perf_counter()
        C, s, dp = o.von_mises(deps.reshape(-1, d), s_h[: sub * nq], p_h[: sub * nq], E=prm.E, nu=prm.nu, sigma_0=prm.sigma_0, H=prm.H, nthreads=nt, out=outb)
        t2 = time.perf_counter()
        o.operand_eps_adjoint(m, s.reshape(sub, nq, d), nn, cells=sub, nthreads=nt)
        t3 = time.perf_counter()
        o.tangent_apply(m, C, v_h, nn, cells=sub, nthreads=nt)
        t4 = time.perf_counter()
        rate = sub * nq / (t4 - t0)
        scan[nt] = rate
        if best is None or rate > best[0]:
            best = (rate, nt, {"operand": t1 - t0, "return_map": t2 - t1, "internal_force": t3 - t2, "tangent_apply": t4 - t3}, sub)
    rate, nt, secs, sub = best
    return {"value": rate, "unit": "qp/s per Newton iteration (1 matvec)", "cores": nt, "kind": "port", "value_1core": scan.get(1), "thread_scan": scan,
            "seconds": secs,
            "sample": f"first {sub} cells ({sub * nq} points) of the same mesh: oracle/operand_oracle_c.c (strain, internal force, tangent action; "
                      f"OpenMP over cells, atomic adds into the dof vector) and oracle/dxo_oracle.c (return map), fastest of the scanned thread counts"}


def assign_leg(torch, ctx, stream, cells_per_side=108, launches=10):
    """dxo_assign on a continuous (CG) dofmap: the Q2 hexahedral dofmap of the device_loop mesh, one scalar value per (cell, local
    node) scattered with NumPy's last-writer-wins rule (external_operator.py:286-287 with get_unrolled_dofmap :18-26).
    Algorithmic bytes: flat_dofs (int32) + values read once, coefficient written once."""
    from dolfinx_external_operator_amd import AssignDesc
    from tools.synthetic import structured_mesh_cached

    dev = torch.device("cuda", ctx.device)
    m = structured_mesh_cached("hexahedron", (cells_per_side,) * 3, 2, distort=0.2, seed=0)     # the device_loop leg's mesh: same dofmap
    nc, npt = m.dofmap.shape
    size = m.node_x.shape[0]
    dofs = torch.from_numpy(np.ascontiguousarray(m.dofmap.reshape(-1))).to(dev)
    vals = torch.randn(nc * npt, dtype=torch.float64, device=dev)
    coeff = torch.zeros(size, dtype=torch.float64, device=dev)
    desc = AssignDesc(nc, npt, 1, 0, npt, 1, 0)
    plan = ctx.assign_plan(desc, dofs.data_ptr(), size)     # first: the counter pass takes the LAST assign_owner dispatches as dxo_assign's
    saved = ctx.get_option("assign_validate")
    res = {}
    try:
        for validate in (1, 0):
            ctx.set_option("assign_validate", validate)
            ms, _ = _time(torch, stream, lambda: ctx.assign(desc, dofs.data_ptr(), vals.data_ptr(), coeff.data_ptr(), size), launches, warm=2)
            res[validate] = ms
    finally:
        ctx.set_option("assign_validate", saved)
    coeff_p = torch.zeros(size, dtype=torch.float64, device=dev)
    ms_plan, _ = _time(torch, stream, lambda: plan.apply(vals.data_ptr(), coeff_p.data_ptr()), launches, warm=2)
    plan_equal = bool(torch.equal(coeff_p, coeff))
    plan_form = plan.form()
    plan.close()
    # NumPy's answer on a sample of dofs: the last (cell, node) entry that targets the dof
    h_d = m.dofmap.reshape(-1)
    probe = np.random.Generator(np.random.PCG64(5)).integers(0, size, 64)
    last = {int(k): int(np.flatnonzero(h_d == k)[-1]) for k in probe[:8]}
    got = coeff.cpu().numpy()
    vh = vals.cpu().numpy()
    ok = all(got[k] == vh[e] for k, e in last.items())
    by = dofs.numel() * 4 + vals.numel() * 8 + size * 8
    ms = res[0]
    return {"workload": f"dxo_assign: CG (Q2 hexahedra, {cells_per_side}^3 cells) dofmap scatter of {nc * npt} values into {size} dofs, "
                        "NumPy last-writer-wins order, fp64", "entries": nc * npt, "dofs": size, "value": nc * npt / ms * 1e3, "unit": "entries/s",
            "ms_per_call": ms, "ms_per_call_with_range_check_sync": res[1], "last_writer_spot_check": "ok" if ok else "MISMATCH",
            "kernels": ["hipMemsetAsync(owner)", "assign_owner", "assign_store"], "dtype": "f64 values / int32 dofs",
            "plan": {"ms_per_call": ms_plan, "equal_to_dxo_assign": plan_equal, "kernels": ["assign_apply_pairs" if plan_form["form"] == 2 else "assign_apply"],
                     "order": {1: "by coefficient entry", 2: "by position in values"}.get(plan_form["form"], "undecided"),
                     "first_apply_ms": {"by_coefficient_entry": plan_form["ms_dof_order"], "by_position_in_values": plan_form["ms_source_order"]},
                     "meaning": "dxo_assign_plan_create once per dofmap, then dxo_assign_apply per call: one gather coeff[d] = values[src[d]], in "
                                "whichever of its two orders the plan's first apply timed faster on these arrays",
                     "roofline": {**_hbm(size * (4 + 8 + 8), ms_plan),
                                  "note": "algorithmic bytes of the planned form: a 4-byte source position, the winning value and the coefficient entry per dof"}},
            "roofline": {**_hbm(by, ms), "algorithmic_bytes_per_call": by,
                         "note": "implementation traffic is higher by construction: a 4-byte owner word per dof is cleared, updated with atomicMax "
                                 "(one per entry: the pass is bound by the atomic rate, ~77 G/s) and read back per entry so that the result is the "
                                 "reference's sequential one, not a race"}}
