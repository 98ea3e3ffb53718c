"""Round-3 additions on the GPU: the reference's 0/0 point carried through the compact (sigma, dp) exchange, the
(sigma, dp)-only form of the von Mises kernel, bit-identical replicas of the compact gather, RCCL buffers refused when
they lie in chunk-backed arena blocks, empty partitions in the resident-state update."""
import numpy as np
import pytest

from conftest import assert_close_scaled, vm_indeterminate_sigma0, vm_inputs
from dolfinx_external_operator_amd import MEM_DEVICE, VmParams
from dolfinx_external_operator_amd._lib import DxoError

pytestmark = pytest.mark.gpu

E, NU = 70e3, 0.3
H = E * (E / 100.0) / (E - E / 100.0)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _kernel_eval(ctx, d):
    """(deps, sigma_n, p, sigma_0) -> (C_tang, sigma, dp) through dxo_von_mises on device memory."""
    import torch

    def run(deps, sigma_n, p, sigma_0, C_null=False):
        n = len(p)
        t = [_dev(a) for a in (deps, sigma_n, p)]
        C = torch.full((n * d * d,), 7.0, dtype=torch.float64, device="cuda:0")
        s = torch.empty(n * d, dtype=torch.float64, device="cuda:0")
        dp = torch.empty(n, dtype=torch.float64, device="cuda:0")
        ctx.von_mises(VmParams(E, NU, sigma_0, H), d, n, MEM_DEVICE, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                      None if C_null else C.data_ptr(), s.data_ptr(), dp.data_ptr())
        ctx.synchronize()
        return C.cpu().numpy().reshape(n, d, d), s.cpu().numpy().reshape(n, d), dp.cpu().numpy()

    return run


def _batch_with_indeterminate_points(ctx, d, n, seed, where):
    run = _kernel_eval(ctx, d)
    sigma_0, sn_row = vm_indeterminate_sigma0(run, d)
    deps, sigma_n, p = vm_inputs(n, d, seed=seed)
    for i in where:
        deps[i], sigma_n[i], p[i] = 0.0, sn_row, 0.0
    return run, sigma_0, deps, sigma_n, p


@pytest.mark.parametrize("d", [4, 6])
def test_indeterminate_point_is_marked_rebuilt_as_nan_and_cleared(ctx, d):
    """f_elastic == 0 exactly: the reference's n_elas is 0/0 and its tangent NaN (demo_plasticity_von_mises.py:318). The
    kernel's own tangent is NaN there; with option vm_mark_indeterminate the point's dp is -0.0, dxo_vm_expand_tangent
    turns the mark into the same NaN tangent, dxo_vm_clear_marks restores +0."""
    import torch

    n, where = 1000, (0, 63, 64, 517, 999)
    run, sigma_0, deps, sigma_n, p = _batch_with_indeterminate_points(ctx, d, n, 31, where)
    prm = VmParams(E, NU, sigma_0, H)
    C0, s0, dp0 = run(deps, sigma_n, p, sigma_0)
    for i in where:
        assert np.isnan(C0[i]).all() and np.isfinite(s0[i]).all() and dp0[i] == 0.0 and not np.signbit(dp0[i])
    assert not np.signbit(dp0).any()
    ctx.set_option("vm_mark_indeterminate", 1)
    try:
        C1, s1, dp1 = run(deps, sigma_n, p, sigma_0)
        _, s2, dp2 = run(deps, sigma_n, p, sigma_0, C_null=True)      # the (sigma, dp)-only kernel carries the same mark
    finally:
        ctx.set_option("vm_mark_indeterminate", 0)
    assert np.array_equal(C1, C0, equal_nan=True) and np.array_equal(s1, s0) and np.array_equal(dp1, dp0)   # -0.0 == 0.0
    assert sorted(np.flatnonzero(np.signbit(dp1))) == sorted(where)
    assert np.array_equal(s2, s0) and np.array_equal(dp2, dp1) and np.array_equal(np.signbit(dp2), np.signbit(dp1))
    # rebuild from the marked state: NaN exactly where the kernel's own tangent is NaN, equal to rounding elsewhere
    for variant in (1, 0):
        ctx.set_option("vm_variant", variant)
        try:
            ts, tdp = _dev(s1.reshape(-1)), _dev(dp1)
            Cx = torch.zeros(n * d * d, dtype=torch.float64, device="cuda:0")
            ctx.vm_expand_tangent(prm, d, n, MEM_DEVICE, ts.data_ptr(), tdp.data_ptr(), Cx.data_ptr())
            ctx.vm_clear_marks(n, tdp.data_ptr())
            ctx.synchronize()
        finally:
            ctx.set_option("vm_variant", 1)
        assert_close_scaled(Cx.cpu().numpy(), C0, 1e-13, f"rebuilt tangent, variant {variant}")
        back = tdp.cpu().numpy()
        assert not np.signbit(back).any() and np.array_equal(back, dp0)
    # without the mark the point leaves no trace in (sigma, dp): it is rebuilt as C_elas (documented)
    ts, tdp = _dev(s0.reshape(-1)), _dev(dp0)
    Cx = torch.zeros(n * d * d, dtype=torch.float64, device="cuda:0")
    ctx.vm_expand_tangent(prm, d, n, MEM_DEVICE, ts.data_ptr(), tdp.data_ptr(), Cx.data_ptr())
    ctx.synchronize()
    assert np.isfinite(Cx.cpu().numpy().reshape(n, d, d)[list(where)]).all()


@pytest.mark.parametrize("n", [1, 64, 1000, 100_001])
def test_state_only_kernel_matches_the_full_kernel(ctx, n, d=6):
    """C_tang = NULL on the device path: (sigma, dp) bit-identical to the full call, the tangent array is never touched;
    on the host path a NULL tangent stays an error."""
    run = _kernel_eval(ctx, d)
    deps, sigma_n, p = vm_inputs(n, d, seed=5)
    C, s, dp = run(deps, sigma_n, p, 250.0)
    Cn, sn, dpn = run(deps, sigma_n, p, 250.0, C_null=True)
    assert np.array_equal(s, sn) and np.array_equal(dp, dpn)
    assert np.all(Cn == 7.0) and not np.all(C == 7.0)
    for variant in (0,):
        ctx.set_option("vm_variant", variant)
        try:
            _, s0, dp0 = run(deps, sigma_n, p, 250.0, C_null=True)
        finally:
            ctx.set_option("vm_variant", 1)
        assert_close_scaled(s0, s, 1e-13, "sigma, lane-per-point kernel")
    from dolfinx_external_operator_amd import MEM_HOST

    out_s, out_dp = np.empty(n * d), np.empty(n)
    with pytest.raises((DxoError, ValueError)):
        ctx.von_mises(VmParams(E, NU, 250.0, H), d, n, MEM_HOST, deps, sigma_n, p, None, out_s, out_dp)


@pytest.mark.parametrize("form", ["single_process", "rank"])
def test_mgpu_compact_reproduces_the_nan_tangent_and_clears_the_marks(ctx, form, d=6):
    """dxo_mgpu_von_mises, world of one: COMPACT = (sigma, dp)-only kernel + rebuild of every block (here: the only one) +
    clear marks. Its result equals FULL's to rounding with the same NaN pattern, dp is +0 at the marked points."""
    import torch

    from dolfinx_external_operator_amd import GATHER_COMPACT, GATHER_FULL, MultiGpu

    n, where = 6400, (5, 64, 6399)
    _, sigma_0, deps, sigma_n, p = _batch_with_indeterminate_points(ctx, d, n, 32, where)
    prm = VmParams(E, NU, sigma_0, H)
    g = MultiGpu(devices=[0]) if form == "single_process" else MultiGpu.from_rank(ctx, MultiGpu.unique_id(), 0, 1)
    try:
        g.set_stream(0, torch.cuda.current_stream().cuda_stream)
        t_in = [_dev(a) for a in (deps, sigma_n, p)]
        res = {}
        for gather in (GATHER_FULL, GATHER_COMPACT):
            C = torch.full((n * d * d,), 3.0, dtype=torch.float64, device="cuda:0")
            s = torch.full((n * d,), float("nan"), dtype=torch.float64, device="cuda:0")
            dp = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0")
            g.von_mises(prm, d, n, gather, [t_in[0]], [t_in[1]], [t_in[2]], [C], [s], [dp])
            g.synchronize()
            res[gather] = (C.cpu().numpy().reshape(n, d, d), s.cpu().numpy(), dp.cpu().numpy())
        Cf, sf, dpf = res[GATHER_FULL]
        Cc, sc, dpc = res[GATHER_COMPACT]
        for i in where:
            assert np.isnan(Cf[i]).all() and np.isnan(Cc[i]).all()
        assert_close_scaled(Cc, Cf, 1e-13, "compact vs full tangent (NaN pattern included)")
        assert np.array_equal(sc, sf) and np.array_equal(dpc, dpf)
        assert not np.signbit(dpc).any() and not np.signbit(dpf).any()
        assert int(g.ctx_option(0, "vm_mark_indeterminate")) == 0          # the call restores the caller's option
        assert int(g.ctx_option(0, "placement_vmm")) == 0                  # arena blocks of a group are hipMalloc memory
    finally:
        g.close()


@pytest.mark.parametrize("form", ["single_process", "rank"])
def test_mgpu_direct_and_pipelined_exchange_forms_world_of_one(ctx, form, d=6):
    """DXO_GATHER_COMPACT_DIRECT / _PIPELINED (round 5: the exchange forms sharding.py has, inside libdxo for C / MPI callers). A
    world of one has nobody to talk to, so what runs here is everything around the exchange: the (sigma, dp)-only kernel in pieces
    on 64-point borders (ragged last piece), the library's exchange stream and events, the per-piece rebuild of the tangents and the
    clearing of the marks — the arrays come out bit for bit as DXO_GATHER_COMPACT leaves them, NaN tangents included."""
    import torch

    from dolfinx_external_operator_amd import GATHER_COMPACT, GATHER_COMPACT_DIRECT, GATHER_COMPACT_PIPELINED, MultiGpu

    n, where = 10_000, (5, 64, 2559, 2560, 9999)     # 10 000 = 156 tiles of 64 + 16: pieces of 2 560 with mgpu_chunks = 4, the last one short
    _, sigma_0, deps, sigma_n, p = _batch_with_indeterminate_points(ctx, d, n, 32, where)
    prm = VmParams(E, NU, sigma_0, H)
    mg = MultiGpu(devices=[0]) if form == "single_process" else MultiGpu.from_rank(ctx, MultiGpu.unique_id(), 0, 1)
    try:
        mg.set_stream(0, torch.cuda.current_stream().cuda_stream)
        t_in = [_dev(a) for a in (deps, sigma_n, p)]
        out = {}
        for gather, chunks in ((GATHER_COMPACT, 4), (GATHER_COMPACT_DIRECT, 4), (GATHER_COMPACT_PIPELINED, 4), (GATHER_COMPACT_PIPELINED, 1),
                               (GATHER_COMPACT_PIPELINED, 7)):
            mg.set_option("mgpu_chunks", chunks)
            C = torch.full((n * d * d,), -5.0, dtype=torch.float64, device="cuda:0")
            s = torch.full((n * d,), -5.0, dtype=torch.float64, device="cuda:0")
            dp = torch.full((n,), -5.0, dtype=torch.float64, device="cuda:0")
            mg.von_mises(prm, d, n, gather, [t_in[0]], [t_in[1]], [t_in[2]], [C], [s], [dp])
            mg.synchronize()
            out[(gather, chunks)] = (C, s, dp)
        ref = out[(GATHER_COMPACT, 4)]
        for i in where:
            assert bool(torch.isnan(ref[0][i * d * d:(i + 1) * d * d]).all())
        assert not bool(torch.signbit(ref[2]).any())
        for key, (C, s, dp) in out.items():
            assert torch.equal(C.view(torch.int64), ref[0].view(torch.int64)), key     # NaN-safe bit comparison
            assert torch.equal(s, ref[1]) and torch.equal(dp.view(torch.int64), ref[2].view(torch.int64)), key
        with pytest.raises(ValueError):
            mg.von_mises(prm, d, n, 5, [t_in[0]], [t_in[1]], [t_in[2]], [ref[0]], [ref[1]], [ref[2]])      # no such gather mode
    finally:
        mg.set_option("mgpu_chunks", 4)
        mg.close()


def test_collectives_refuse_chunk_backed_arena_blocks(hip_library):
    """A virtual range backed by 2 MB physical chunks is accessible from its own device only and cannot be exported to a
    peer: dxo_mgpu_all_gather returns DXO_E_MEM for a pointer inside such an arena block instead of handing it to RCCL."""
    import torch

    from dolfinx_external_operator_amd import MultiGpu

    g = MultiGpu(devices=[0])
    try:
        g.set_stream(0, torch.cuda.current_stream().cuda_stream)
        g.set_option("placement_vmm", 2)            # the user overrides the group's default: chunk-backed candidates only
        g.set_option("placement_candidates", 2)
        g.set_option("placement_min_bytes", 1 << 22)
        c = g.context(0)
        (buf,) = c.output_tensors([1 << 20])         # 8 MiB > placement_min_bytes: calibrated, chunk-backed
        info = buf.dxo_block.info
        assert info["mode"] == "candidates" and info["chosen_kind"] == "2MB_chunks"
        with pytest.raises(ValueError, match="chunk"):
            g.all_gather([buf], 1 << 20)
        plain = torch.zeros(1024, dtype=torch.float64, device="cuda:0")
        g.all_gather([plain], 1024)
        g.synchronize()
        del buf
    finally:
        g.close()


def test_resident_state_commit_on_an_empty_partition(ctx):
    """A rank without cells (the reference handles an empty partition, external_operator.py:365-371): upload, call and
    the load-step update are no-ops, not errors."""
    st = ctx.vm_state(6, 0)
    empty = np.empty(0)
    st.upload(empty, empty)
    st.call(VmParams(E, NU, 250.0, H), MEM_DEVICE, None, None)
    st.commit()
    st.commit()
    st.close()


def test_bind_puts_the_results_into_the_operators_coefficient(ctx, oracle):
    """`external_function.bind(operator, sigma_holder, dp_holder)`: the one-line form of `outputs=`. Element 0 of the result
    IS the operator's coefficient storage, so the reference's `coefficient.x.array[:] = values` (external_operator.py:441)
    is an assignment of an array to itself; the extras land in the holders the demo copies them into
    (demo_plasticity_von_mises.py:451-456)."""
    from dolfinx_external_operator_amd import (Coefficient, QuadratureExternalOperator, evaluate_external_operators, evaluate_operands,
                                               make_mohr_coulomb, make_von_mises)
    from dolfinx_external_operator_amd.evaluation import Operand

    nc, nq, d = 500, 8, 6
    n = nc * nq
    deps, sigma_n, p = vm_inputs(n, d, seed=41)
    op = QuadratureExternalOperator(Operand(lambda cells: deps.reshape(nc, nq, d)[cells], "deps"), num_cells=nc, num_points=nq,
                                    value_shape=(d, d), derivatives=(1,))
    sigma_new, dp_new = Coefficient(n * d), Coefficient(n)
    op.external_function = make_von_mises(sigma_n, p, ctx=ctx)
    assert op.external_function.bind(op, sigma_new, dp_new) is op.external_function
    ((C, s, dpv),) = evaluate_external_operators([op], evaluate_operands([op]))
    assert np.shares_memory(C, op.ref_coefficient.x.array) and np.shares_memory(s, sigma_new.x.array) and np.shares_memory(dpv, dp_new.x.array)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(op.ref_coefficient.x.array, Co, 1e-13, "coefficient")
    assert_close_scaled(sigma_new.x.array, so, 1e-13, "sigma")
    assert_close_scaled(dp_new.x.array, dpo, 1e-13, "dp")
    with pytest.raises(ValueError):
        op.external_function.bind(op, sigma_new, dp_new, dp_new)
    op.external_function.bind()                   # unbound again: fresh arrays
    ((C2, _, _),) = evaluate_external_operators([op], evaluate_operands([op]))
    assert not np.shares_memory(C2, op.ref_coefficient.x.array) and np.array_equal(C2, C)
    mc = make_mohr_coulomb(np.zeros((10, 4)), ctx=ctx)
    tgt = Coefficient(160)
    mc.bind(tgt)
    Cm, _ = mc((1,))(np.zeros((10, 4)))
    assert np.shares_memory(Cm, tgt.x.array)


def test_device_calls_keep_their_outputs_in_a_persistent_arena_block(ctx, oracle):
    """CUDA-tensor operands, device_outputs="arena" (opt-in): above the arena threshold the outputs live in ONE calibrated block
    that the next call overwrites (documented aliasing); below it, and with the default device_outputs="fresh", every call
    returns new tensors — the reference's callbacks return new arrays, so results of call k survive call k+1 by default.
    Operand device / dtype are checked before anything is allocated or calibrated."""
    import torch

    from dolfinx_external_operator_amd import make_von_mises

    d, nq = 6, 8
    saved = ctx.get_option("placement_min_bytes"), ctx.get_option("placement_candidates")
    ctx.set_option("placement_min_bytes", 1 << 22)     # 4 MiB so that a test-sized batch is "large"
    ctx.set_option("placement_candidates", 3)
    try:
        n = 64_000
        deps, sigma_n, p = vm_inputs(n, d, seed=42)
        t = [_dev(a) for a in (deps.reshape(n // nq, nq, d), sigma_n.reshape(-1), p)]
        ext = make_von_mises(t[1], t[2], ctx=ctx, device_outputs="arena")
        with pytest.raises(TypeError):
            ext((1,))(t[0].float())                                  # refused before the arena block is made
        C1, s1, dp1 = ext((1,))(t[0])
        ptr = C1.data_ptr()
        assert C1.dxo_block.info["mode"] == "candidates"
        Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
        assert_close_scaled(C1.cpu().numpy(), Co, 1e-13, "arena-backed device call")
        C2, _, _ = ext((1,))(t[0] * 0.5)
        assert C2.data_ptr() == ptr                                  # the same block: the first result has been overwritten
        assert_close_scaled(C1.cpu().numpy(), oracle.von_mises(deps * 0.5, sigma_n, p)[0], 1e-13, "aliased view sees the new call")
        small = [_dev(a) for a in (deps[:800].reshape(100, nq, d), sigma_n[:800].reshape(-1), p[:800])]
        ext_s = make_von_mises(small[1], small[2], ctx=ctx, device_outputs="arena")
        a = ext_s((1,))(small[0])[0]
        b = ext_s((1,))(small[0])[0]
        assert a.data_ptr() != b.data_ptr()                          # below the threshold: fresh tensors
        fresh = make_von_mises(t[1], t[2], ctx=ctx)                  # the default
        x = fresh((1,))(t[0])[0]
        y = fresh((1,))(t[0] * 0.5)[0]
        assert x.data_ptr() != y.data_ptr() and not hasattr(x, "dxo_block")
        assert_close_scaled(x.cpu().numpy(), Co, 1e-13, "the first result survives the second call")
    finally:
        ctx.set_option("placement_min_bytes", saved[0])
        ctx.set_option("placement_candidates", saved[1])


def test_isihara_device_calls_use_a_persistent_arena_block(ctx):
    """The HBM-bound analytic operator, CUDA-tensor operand, device_outputs="arena" (opt-in): above the arena threshold (dP, P)
    live in one block chosen by timing the kernel itself, overwritten by the next call; results equal the default
    fresh-tensor form bit for bit."""
    import torch

    from dolfinx_external_operator_amd import make_isihara

    saved = ctx.get_option("placement_min_bytes"), ctx.get_option("placement_candidates")
    ctx.set_option("placement_min_bytes", 1 << 22)
    ctx.set_option("placement_candidates", 3)
    try:
        n = 60_000
        g = torch.Generator(device="cuda:0").manual_seed(9)
        F = torch.randn(n, 1, 2, 2, device="cuda:0", dtype=torch.float64, generator=g) * 0.05 + torch.eye(2, device="cuda:0", dtype=torch.float64)
        ext, fresh = make_isihara(ctx=ctx, device_outputs="arena"), make_isihara(ctx=ctx)
        dP1, P1 = ext((1,))(F)
        assert dP1.dxo_block.info["mode"] == "candidates" and dP1.numel() == n * 16 and P1.numel() == n * 4
        dPf, Pf = fresh((1,))(F)
        assert torch.equal(dP1, dPf) and torch.equal(P1, Pf) and not hasattr(dPf, "dxo_block")
        ptr = dP1.data_ptr()
        dP2, _ = ext((1,))(F * 1.01)
        assert dP2.data_ptr() == ptr and torch.equal(dP1, fresh((1,))(F * 1.01)[0])      # aliased: the first view sees the new call
        small = ext((1,))(F[:100])[0], ext((1,))(F[:100])[0]
        assert small[0].data_ptr() != small[1].data_ptr()                                 # below the threshold: fresh tensors
        with pytest.raises(ValueError):
            make_isihara(ctx=ctx, device_outputs="pinned")
    finally:
        ctx.set_option("placement_min_bytes", saved[0])
        ctx.set_option("placement_candidates", saved[1])


@pytest.mark.parametrize("form", ["single_process", "rank"])
def test_mgpu_sharded_entry_points_of_the_other_operators(ctx, oracle, golden, form):
    """dxo_mgpu_mohr_coulomb / _icnn / _isihara / _heat with a world of one: the communicator comes up, every operator
    writes its block into the full-length outputs, DXO_GATHER_FULL is the in-place collective, results are those of the
    single-GPU entry points bit for bit; optional outputs may be left out; DXO_GATHER_COMPACT is refused (it exists for
    von Mises only). World > 1 differs by block offsets only (tests/test_sharding.py covers that arithmetic)."""
    import torch

    from conftest import mc_tracing_inputs
    from dolfinx_external_operator_amd import GATHER_COMPACT, GATHER_FULL, GATHER_NONE, IsiharaParams, MultiGpu
    from tools.mc_inputs import mc_default_params

    g = MultiGpu(devices=[0]) if form == "single_process" else MultiGpu.from_rank(ctx, MultiGpu.unique_id(), 0, 1)
    c = g.context(0)
    model = None
    try:
        g.set_stream(0, torch.cuda.current_stream().cuda_stream)
        c.set_stream(torch.cuda.current_stream().cuda_stream)
        f64 = dict(dtype=torch.float64, device="cuda:0")
        n = 4096
        # ---- Mohr-Coulomb
        prm = mc_default_params()
        deps, sn = mc_tracing_inputs(oracle, n, seed=51)
        td, ts = _dev(deps), _dev(sn)
        ref = [torch.empty(n * 16, **f64), torch.empty(n * 4, **f64), torch.empty(n, dtype=torch.int32, device="cuda:0"), torch.empty(n, **f64)]
        c.mohr_coulomb(prm, n, MEM_DEVICE, td.data_ptr(), ts.data_ptr(), ref[0].data_ptr(), ref[1].data_ptr(), ref[2].data_ptr(), ref[3].data_ptr())
        for gather in (GATHER_NONE, GATHER_FULL):
            out = [torch.zeros_like(t) for t in ref]
            g.mohr_coulomb(prm, n, gather, [td], [ts], [out[0]], [out[1]], niter=[out[2]], yielding=[out[3]])
            g.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(out, ref))
        with pytest.raises(ValueError, match="von Mises only"):
            g.mohr_coulomb(prm, n, GATHER_COMPACT, [td], [ts], [out[0]], [out[1]])
        with pytest.raises(ValueError):
            g.mohr_coulomb(prm, n - 2, GATHER_FULL, [td], [ts], [out[0]], [out[1]])         # blocks must stay 16-byte aligned
        # ---- ICNN + analytic Isihara
        gd, w = golden("icnn_isihara.npz"), golden("icnn_isihara_weights.npz")
        F = _dev(np.tile(gd["F"].reshape(-1, 4), (3, 1))[:n])
        model = c.icnn_create({k.replace("__", "."): w[k] for k in w.files})
        r_dP, r_P = torch.empty(n * 16, **f64), torch.empty(n * 4, **f64)
        c.icnn_eval(model, 0, n, MEM_DEVICE, F.data_ptr(), r_dP.data_ptr(), r_P.data_ptr())
        dP, P = torch.zeros(n * 16, **f64), torch.zeros(n * 4, **f64)
        g.icnn([model], 0, n, GATHER_FULL, [F], [dP], [P])
        g.synchronize()
        assert torch.equal(dP, r_dP) and torch.equal(P, r_P)
        ip = IsiharaParams(0.5, 1.0, 1.0, 1.5)
        c.isihara(ip, n, MEM_DEVICE, F.data_ptr(), r_dP.data_ptr(), r_P.data_ptr())
        g.isihara(ip, n, GATHER_FULL, [F], [dP], [P])
        g.synchronize()
        assert torch.equal(dP, r_dP) and torch.equal(P, r_P)
        # ---- heat (only two of the three outputs requested)
        h = golden("heat_c1.npz")
        T, sg = _dev(h["T"].reshape(-1)[:n]), _dev(h["sigma"].reshape(-1, 2)[:n])
        q, ds = torch.zeros(n * 2, **f64), torch.zeros(n * 4, **f64)
        g.heat(1.0, 1.0, 2, n, GATHER_FULL, [T], [sg], q=[q], dqdsigma=[ds])
        g.synchronize()
        assert np.array_equal(q.cpu().numpy(), h["q"][: n * 2]) and np.array_equal(ds.cpu().numpy(), h["dqdsigma"][: n * 4])
    finally:
        if model is not None:
            c.icnn_destroy(model)
        g.close()


@pytest.mark.parametrize("n", [1, 63, 600, 6144])
def test_tiny_host_batches_without_dma_are_bit_identical(ctx, n, d=4):
    """At the reference's demo sizes a host call no longer copies by DMA: the kernel reads / writes the page-locked staging
    block over PCIe itself (option host_zero_copy_bytes). Same bytes as the copy form, the chunked pipeline and a device
    call; guard words behind the outputs stay untouched."""
    from dolfinx_external_operator_amd import MEM_HOST

    deps, sigma_n, p = vm_inputs(n, d, seed=60 + n)
    prm = VmParams(E, NU, 250.0, H)
    res = {}
    saved = ctx.get_option("host_zero_copy_bytes"), ctx.get_option("host_small_bytes")
    try:
        for name, zc, small in (("zero_copy", 1 << 22, 1 << 22), ("copy", 0, 1 << 22), ("chunked", 0, 0)):
            ctx.set_option("host_zero_copy_bytes", zc)
            ctx.set_option("host_small_bytes", small)
            C, s, dp = np.full(n * d * d + 3, -7.0), np.full(n * d + 3, -7.0), np.full(n + 3, -7.0)
            ctx.von_mises(prm, d, n, MEM_HOST, deps, sigma_n, p, C, s, dp)
            assert np.all(C[n * d * d:] == -7.0) and np.all(s[n * d:] == -7.0) and np.all(dp[n:] == -7.0)
            res[name] = (C.copy(), s.copy(), dp.copy())
    finally:
        ctx.set_option("host_zero_copy_bytes", saved[0])
        ctx.set_option("host_small_bytes", saved[1])
    for name in ("copy", "chunked"):
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(res["zero_copy"], res[name])), name
