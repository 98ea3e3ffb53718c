"""Round-4 changes on the GPU: MultiGpu.context() knows its device and dies with the group, tiny host batches skip the DMA
copies only for the operators whose kernels stream every byte once, the (sigma, dp)-only von Mises launch honours the
`nontemporal` option."""
import numpy as np
import pytest

from conftest import mc_tracing_inputs, vm_inputs
from dolfinx_external_operator_amd import MEM_DEVICE, MEM_HOST, MultiGpu, VmParams

pytestmark = pytest.mark.gpu

E, NU = 70e3, 0.3
H = E * (E / 100.0) / (E - E / 100.0)


def test_mgpu_context_knows_its_device_and_dies_with_the_group(ctx):
    import torch

    assert ctx.get_option("device") == 0
    with pytest.raises(ValueError):
        ctx.set_option("device", 1)                         # read-only
    g = MultiGpu(devices=[0])
    c = g.context(0)
    assert c is g.context(0) and c.device == g.ctx_option(0, "device") == 0
    blk = c.output_tensors((1024,))[0]                     # an arena block made through the borrowed wrapper
    assert blk.device.index == c.device
    st = c.vm_state(4, 128)
    st.upload(np.zeros(512), np.zeros(128))
    g.close()
    assert c._h is None                                     # no dangling handle: later finalizers find a closed context
    del blk
    st.close()                                              # frees its device block without touching the destroyed dxo_ctx
    torch.cuda.synchronize()
    # the group can be made again afterwards
    g2 = MultiGpu(devices=[0])
    assert g2.context(0).device == 0
    g2.close()


def test_zero_copy_applies_to_stream_once_operators_only(ctx, oracle):
    """Near the 2 MiB small-path boundary: von Mises (stream-once kernel) may run on the device-mapped staging block, the
    Mohr-Coulomb kernel (re-reads its inputs, stores partial lines) keeps the DMA copies; either way results do not depend on
    the option."""
    from tools.mc_inputs import mc_default_params

    saved = ctx.get_option("host_zero_copy_bytes")
    try:
        n = 7000                                            # 64 B of inputs + 164 B of outputs per point = 1.6 MB: on the small path
        deps, sn = mc_tracing_inputs(oracle, n, seed=9)
        prm = mc_default_params()
        res = []
        for zc in (saved, 0):
            ctx.set_option("host_zero_copy_bytes", zc)
            C, s = np.empty(n * 16), np.empty(n * 4)
            it = np.empty(n, dtype=np.int32)
            ctx.mohr_coulomb(prm, n, MEM_HOST, deps, sn, C, s, it)
            res.append((C, s, it))
        assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*res))
        d, m = 4, 6000
        e, sg, p = vm_inputs(m, d, seed=3)
        outs = []
        for zc in (saved, 0):
            ctx.set_option("host_zero_copy_bytes", zc)
            C, s, dp = np.empty(m * d * d), np.empty(m * d), np.empty(m)
            ctx.von_mises(VmParams(E, NU, 250.0, H), d, m, MEM_HOST, e, sg, p, C, s, dp)
            outs.append((C, s, dp))
        assert all(np.array_equal(a, b) for a, b in zip(*outs))
    finally:
        ctx.set_option("host_zero_copy_bytes", saved)


@pytest.mark.parametrize("d", [4, 6])
def test_state_only_launch_honours_nontemporal(ctx, d):
    import torch

    n = 10_000
    e, sg, p = vm_inputs(n, d, seed=d)
    t = [torch.from_numpy(a).cuda() for a in (e, sg, p)]
    prm = VmParams(E, NU, 250.0, H)
    saved = ctx.get_option("nontemporal")
    got = []
    try:
        for nt in (1, 0):
            ctx.set_option("nontemporal", nt)
            s = torch.empty(n * d, dtype=torch.float64, device="cuda")
            dp = torch.empty(n, dtype=torch.float64, device="cuda")
            ctx.von_mises(prm, d, n, MEM_DEVICE, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, s.data_ptr(), dp.data_ptr())
            ctx.synchronize()
            got.append((s.cpu(), dp.cpu()))
    finally:
        ctx.set_option("nontemporal", saved)
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


@pytest.mark.parametrize("cell,n,degree", [("hexahedron", (5, 4, 3), 2), ("hexahedron", (2, 2, 2), 2), ("hexahedron", (3, 3, 2), 1),
                                           ("triangle", (7, 6), 2), ("tetrahedron", (2, 3, 2), 2)])
def test_residual_in_one_call_equals_field_then_adjoint(ctx, cell, n, degree):
    """dxo_von_mises_residual = dxo_von_mises_field (no tangent) followed by dxo_operand_adjoint of the returned stress: bit-identical.
    With option vm_residual_fused = 1 one kernel does both on Q2 hexahedra (the stress is scattered from registers): (sigma, dp)
    bit-identical, R to rounding of the Jacobian's sum order; on every other element the option changes nothing."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh
    from tools.synthetic import structured_mesh

    m = structured_mesh(cell, n, degree, distort=0.2, seed=4)
    G, d = m.gdim, 4 if m.gdim == 2 else 6
    nn, npts = m.node_x.shape[0], m.num_cells * m.nq
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    prm = VmParams(E, NU, 250.0, H)
    rng = np.random.Generator(np.random.PCG64(11))
    f64 = dict(dtype=torch.float64, device="cuda")
    u = torch.from_numpy(rng.normal(0.0, 4e-3, size=nn * G)).cuda()
    sn = torch.from_numpy(rng.normal(0.0, 100.0, size=npts * d)).cuda()
    p = torch.from_numpy(np.abs(rng.normal(0.0, 1e-3, size=npts))).cuda()
    R0 = torch.from_numpy(rng.normal(size=nn * G)).cuda()            # the call ADDS to R
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        sig_a, dp_a, R_a = torch.zeros(npts * d, **f64), torch.zeros(npts, **f64), R0.clone()
        dm.von_mises(prm, u.data_ptr(), sn.data_ptr(), p.data_ptr(), None, sig_a.data_ptr(), dp_a.data_ptr(), mem=MEM_DEVICE)
        dm.adjoint("eps", G, sig_a.data_ptr(), R_a.data_ptr())
        results = {}
        for fused in (1, 0):
            ctx.set_option("vm_residual_fused", fused)
            sig, dpo, R = torch.full((npts * d,), np.nan, **f64), torch.full((npts,), np.nan, **f64), R0.clone()
            dm.von_mises_residual(prm, u.data_ptr(), sn.data_ptr(), p.data_ptr(), sig.data_ptr(), dpo.data_ptr(), R.data_ptr())
            torch.cuda.synchronize()
            assert torch.equal(sig, sig_a) and torch.equal(dpo, dp_a), (cell, fused)
            results[fused] = R
        assert float((dp_a > 0).double().mean()) > 0.2                   # plastic and elastic points both present
        assert torch.equal(results[0], R_a)
        scale = float((R_a - R0).abs().max())
        assert float((results[1] - R_a).abs().max()) <= 1e-12 * scale
        if not (cell == "hexahedron" and degree == 2):
            assert torch.equal(results[1], R_a)                             # no fused kernel for this element: the same two calls
        again = R0.clone()
        ctx.set_option("vm_residual_fused", 1)
        dm.von_mises_residual(prm, u.data_ptr(), sn.data_ptr(), p.data_ptr(), sig.data_ptr(), dpo.data_ptr(), again.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(again, results[1])                               # no atomics: bit-reproducible
    finally:
        ctx.set_option("vm_residual_fused", 0)
        dm.close()


@pytest.mark.parametrize("cell,n", [("hexahedron", (4, 3, 3)), ("triangle", (7, 6))])
def test_consumer_overwrite_sets_instead_of_adding(ctx, cell, n):
    """Option consumer_overwrite = 1: the consumer-side calls SET `out` (no memset needed before a Krylov matvec): same bits as
    accumulating into zeros, for the two-pass form, the atomics form and an entity subset (whose other dofs become zero)."""
    import torch

    from dolfinx_external_operator_amd import DeviceMesh
    from tools.synthetic import structured_mesh

    m = structured_mesh(cell, n, 2, distort=0.2, seed=4)
    G, d = m.gdim, 4 if m.gdim == 2 else 6
    nn, npts = m.node_x.shape[0], m.num_cells * m.nq
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    prm = VmParams(E, NU, 250.0, H)
    rng = np.random.Generator(np.random.PCG64(12))
    S = torch.from_numpy(rng.normal(0.0, 100.0, size=npts * d)).cuda()
    dpv = torch.from_numpy(np.abs(rng.normal(0.0, 1e-3, size=npts)) * (rng.random(npts) < 0.5)).cuda()
    A = rng.normal(size=(npts, d, d))
    Ct = torch.from_numpy((A @ A.transpose(0, 2, 1) + np.eye(d)).reshape(-1)).cuda()
    v = torch.from_numpy(rng.normal(size=nn * G)).cuda()
    sub = torch.from_numpy(np.arange(0, m.num_cells, 3, dtype=np.int32)).cuda()
    S_sub = S.reshape(m.num_cells, -1)[::3].contiguous().reshape(-1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    calls = {"force": lambda o: dm.adjoint("eps", G, S.data_ptr(), o.data_ptr()),
             "force_subset": lambda o: dm.adjoint("eps", G, S_sub.data_ptr(), o.data_ptr(), n_cells=int(sub.numel()), cells_ptr=sub.data_ptr()),
             "apply": lambda o: dm.tangent_apply(Ct.data_ptr(), v.data_ptr(), o.data_ptr()),
             "diag": lambda o: dm.tangent_diagonal(Ct.data_ptr(), o.data_ptr()),
             "apply_vm": lambda o: dm.tangent_apply_vm(prm, S.data_ptr(), dpv.data_ptr(), v.data_ptr(), o.data_ptr()),
             "diag_vm": lambda o: dm.tangent_diagonal_vm(prm, S.data_ptr(), dpv.data_ptr(), o.data_ptr())}
    try:
        for atomics in (0, 1):
            ctx.set_option("adjoint_atomics", atomics)
            for name, fn in calls.items():
                ctx.set_option("consumer_overwrite", 0)
                ref = torch.zeros(nn * G, dtype=torch.float64, device="cuda")
                fn(ref)
                twice = ref.clone()
                fn(twice)                                                  # default: accumulates
                ctx.set_option("consumer_overwrite", 1)
                out = torch.from_numpy(rng.normal(size=nn * G)).cuda()   # garbage that must not survive
                fn(out)
                torch.cuda.synchronize()
                scale = float(ref.abs().max())
                assert float((twice - 2 * ref).abs().max()) <= 1e-12 * scale, (name, atomics)
                if atomics == 0 and name != "force_subset":
                    assert torch.equal(out, ref), (name, atomics)           # two-pass form: the same bits
                else:
                    assert float((out - ref).abs().max()) <= 1e-12 * scale, (name, atomics)
    finally:
        ctx.set_option("consumer_overwrite", 0)
        ctx.set_option("adjoint_atomics", 0)
        dm.close()

