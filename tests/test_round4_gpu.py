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
