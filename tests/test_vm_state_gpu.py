"""Device-resident von Mises history variables: dxo_vm_state_*, dxo_von_mises_state,
dxo_von_mises_field_state and make_von_mises(state="resident").

The reference re-reads sigma_n and p at every call (demo_plasticity_von_mises.py:347-348) and changes them at the end of
a load step (:564-565). With the mirror they cross PCIe once; every result must stay BIT-identical to the plain entry
points on the same values, and the device-side commit must reproduce the reference's two NumPy statements exactly."""
import warnings

import numpy as np
import pytest

from conftest import assert_close_scaled, vm_inputs
from dolfinx_external_operator_amd import MEM_DEVICE, MEM_HOST, DeviceMesh, VmParams, make_von_mises
from tools.synthetic import structured_mesh

pytestmark = pytest.mark.gpu

E, NU, SIGMA_0 = 70e3, 0.3, 250.0
H = E * (E / 100.0) / (E - E / 100.0)
PRM = VmParams(E, NU, SIGMA_0, H)


def _plain(ctx, d, n, deps, sigma_n, p, rebuild):
    C, s, dp = np.empty(n * d * d), np.empty(n * d), np.empty(n)
    ctx.set_option("vm_host_tangent", int(rebuild))
    try:
        ctx.von_mises(PRM, d, n, MEM_HOST, deps, sigma_n, p, C, s, dp)
    finally:
        ctx.set_option("vm_host_tangent", 0)
    return C, s, dp


@pytest.mark.parametrize("d", [4, 6])
@pytest.mark.parametrize("n", [0, 1, 777, 70_001, 300_037])      # packed small path; chunked; above the rebuild threshold
@pytest.mark.parametrize("rebuild", [0, 1])
def test_state_call_is_bit_identical_to_the_plain_call(ctx, d, n, rebuild):
    deps, sigma_n, p = vm_inputs(max(n, 1), d, seed=5)
    deps, sigma_n, p = deps[:n], sigma_n[:n], p[:n]
    ctx.set_option("host_chunk_points", 1 << 16)
    try:
        ref = _plain(ctx, d, n, deps, sigma_n, p, rebuild)
        st = ctx.vm_state(d, n)
        st.upload(sigma_n, p)
        C, s, dp = np.full(n * d * d, np.nan), np.full(n * d, np.nan), np.full(n, np.nan)
        ctx.set_option("vm_host_tangent", rebuild)
        st.call(PRM, MEM_HOST, deps, C, s, dp)
        ctx.set_option("vm_host_tangent", 0)
    finally:
        ctx.set_option("host_chunk_points", 1 << 20)
        ctx.set_option("vm_host_tangent", 0)
    for got, want, name in zip((C, s, dp), ref, ("C_tang", "sigma", "dp")):
        np.testing.assert_array_equal(got, want, err_msg=name)
    if n:
        # the reference's load-step update, applied to the mirror on the device and to the arrays on the host
        st.commit()
        got_s, got_p = st.download()
        np.testing.assert_array_equal(got_s, s)                       # sigma_n[:] = sigma          (:565)
        np.testing.assert_array_equal(got_p, p + dp)                  # p += dp                      (:564)
        with pytest.raises(ValueError, match="nothing to commit"):
            st.commit()
    st.close()


def test_state_with_device_operands_and_pointers(ctx):
    import torch

    n, d = 50_000, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=6)
    ref = _plain(ctx, d, n, deps, sigma_n, p, 0)
    st = ctx.vm_state(d, n)
    with pytest.raises(ValueError, match="upload"):
        st.call(PRM, MEM_HOST, deps, np.empty(n * d * d), np.empty(n * d), np.empty(n))
    d_sn, d_p = torch.from_numpy(sigma_n).cuda(), torch.from_numpy(p).cuda()
    st.upload(d_sn.data_ptr(), d_p.data_ptr(), MEM_DEVICE)             # device -> mirror
    d_deps = torch.from_numpy(deps).cuda()
    C = torch.empty(n * d * d, dtype=torch.float64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    st.call(PRM, MEM_DEVICE, d_deps.data_ptr(), C.data_ptr())         # sigma, dp stay in the mirror
    torch.cuda.synchronize()
    ptr = st.pointers()
    s, dp = np.empty(n * d), np.empty(n)
    ctx.copy(s, ptr["sigma"], s.nbytes, 1)
    ctx.copy(dp, ptr["dp"], dp.nbytes, 1)
    np.testing.assert_array_equal(C.cpu().numpy(), ref[0])
    np.testing.assert_array_equal(s, ref[1])
    np.testing.assert_array_equal(dp, ref[2])
    s2 = torch.empty(n * d, dtype=torch.float64, device="cuda")
    dp2 = torch.empty(n, dtype=torch.float64, device="cuda")
    st.call(PRM, MEM_DEVICE, d_deps.data_ptr(), C.data_ptr(), s2.data_ptr(), dp2.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(s2.cpu().numpy(), ref[1])
    np.testing.assert_array_equal(dp2.cpu().numpy(), ref[2])
    st.close()


@pytest.mark.parametrize("host_tangent", ["copy", "rebuild"])
def test_resident_factory_follows_a_load_history(ctx, oracle, host_tangent):
    """Five load steps, three calls each (Newton iterations re-evaluate with the same state), against the same factory
    with state='host' (bit-identical) and the oracle; the caller applies the reference's update to its arrays and tells
    the operator with commit_state()."""
    n, d, nq = 90_000, 6, 8                                            # above vm_rebuild_min_points? no: exercised below too
    deps0, sigma_n, p = vm_inputs(n, d, seed=8)
    sigma_n *= 0.0
    p *= 0.0
    sn_b, p_b = sigma_n.copy(), p.copy()
    ext_r = make_von_mises(sigma_n, p, ctx=ctx, state="resident", host_tangent=host_tangent)
    ext_h = make_von_mises(sn_b, p_b, ctx=ctx, host_tangent=host_tangent)
    with pytest.raises(RuntimeError, match="resident"):
        ext_h.commit_state()
    for step in range(5):
        for it in range(3):
            deps = (deps0 * (0.4 + 0.3 * step + 0.01 * it) * (-1.0 if step == 3 else 1.0)).reshape(n // nq, nq, d)
            with warnings.catch_warnings():
                warnings.simplefilter("error")                        # no tripwire warning on the documented protocol
                C_r, s_r, dp_r = ext_r((1,))(deps)
            C_h, s_h, dp_h = ext_h((1,))(deps)
            np.testing.assert_array_equal(C_r, C_h)
            np.testing.assert_array_equal(s_r, s_h)
            np.testing.assert_array_equal(dp_r, dp_h)
        C_o, s_o, dp_o = oracle.von_mises(deps.reshape(n, d), sigma_n, p)
        assert_close_scaled(C_r, C_o.reshape(-1), 1e-13, f"C_tang at step {step}")
        p += dp_r                                                     # :564
        sigma_n[:] = s_r.reshape(n, d)                                # :565
        ext_r.commit_state()
        p_b += dp_h
        sn_b[:] = s_h.reshape(n, d)
        assert ext_r.check_state() == 0.0
    assert (p > 0).mean() > 0.5


def test_resident_factory_notices_unannounced_changes(ctx):
    n, d = 40_000, 4
    deps, sigma_n, p = vm_inputs(n, d, seed=9)
    ext = make_von_mises(sigma_n, p, ctx=ctx, state="resident")
    ref = make_von_mises(sigma_n, p, ctx=ctx)
    a = ext((1,))(deps.reshape(-1, 4, d))
    sigma_n *= 0.5                                                    # no commit_state(), no state_changed()
    with pytest.warns(RuntimeWarning, match="re-uploading"):
        b = ext((1,))(deps.reshape(-1, 4, d))
    for x, y in zip(b, ref((1,))(deps.reshape(-1, 4, d))):
        np.testing.assert_array_equal(x, y)
    assert not np.array_equal(a[1], b[1])
    p[3] += 1.0                                                       # a change the sampled tripwire may miss: announced
    ext.state_changed()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        c = ext((1,))(deps.reshape(-1, 4, d))
    for x, y in zip(c, ref((1,))(deps.reshape(-1, 4, d))):
        np.testing.assert_array_equal(x, y)
    assert ext.check_state() == 0.0


@pytest.mark.parametrize("cell,n", [("triangle", (150, 120)), ("hexahedron", (34, 31, 32))])
@pytest.mark.parametrize("host_tangent", ["copy", "rebuild"])
def test_lazy_operand_with_resident_state(ctx, cell, n, host_tangent):
    """evaluate_operands -> LazyOperand -> make_von_mises(state='resident'): only the dof vector goes up."""
    m = structured_mesh(cell, n, 2, distort=0.15, seed=1)
    dm = DeviceMesh.from_synthetic(m, ctx=ctx)
    d = 4 if m.gdim == 2 else 6
    npts = m.num_cells * m.nq
    rng = np.random.Generator(np.random.PCG64(4))
    u = rng.normal(0.0, 2e-3, m.node_x.shape[0] * m.gdim)
    sigma_n = rng.normal(0.0, 60.0, (npts, d))
    p = np.abs(rng.normal(0.0, 1e-3, npts))
    ext_r = make_von_mises(sigma_n, p, ctx=ctx, state="resident", host_tangent=host_tangent)
    ext_h = make_von_mises(sigma_n, p, ctx=ctx, host_tangent="copy")
    for step in range(2):
        got = ext_r((1,))(dm.operand("eps", u * (1.0 + step), lazy=True).eval(None))
        want = ext_h((1,))(dm.operand("eps", u * (1.0 + step), lazy=True).eval(None))
        np.testing.assert_array_equal(got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
        if host_tangent == "copy":
            np.testing.assert_array_equal(got[0], want[0])
        else:
            assert_close_scaled(got[0], want[0], 1e-14, "rebuilt tangent")
        p += got[2]
        sigma_n[:] = got[1].reshape(npts, d)
        ext_r.commit_state()
        assert ext_r.check_state() == 0.0


# ---------------------------------------------------------------------------------------------------------------------
# NumPy arrays sharded over several pipelines without a collective (dxo_mgpu_create_local + dxo_mgpu_von_mises_host).
# A 1-GPU box lists device 0 several times: every entry is its own context, streams and thread, so the split, the
# concurrency and the thread-budget sharing are the real thing; only the extra PCIe links are missing.
@pytest.mark.parametrize("n_dev", [1, 2, 3])
@pytest.mark.parametrize("n", [0, 5, 100_003, 700_001])
@pytest.mark.parametrize("rebuild", [0, 1])
def test_host_arrays_sharded_over_local_pipelines(ctx, n_dev, n, rebuild):
    from dolfinx_external_operator_amd._lib import MultiGpu

    d = 6
    deps, sigma_n, p = vm_inputs(max(n, 1), d, seed=12)
    deps, sigma_n, p = deps[:n], sigma_n[:n], p[:n]
    ref = _plain(ctx, d, n, deps, sigma_n, p, rebuild)
    g = MultiGpu.local([0] * n_dev)
    try:
        assert g.local_count == n_dev and g.world == n_dev
        g.set_option("vm_host_tangent", rebuild)
        g.set_option("vm_rebuild_min_points", 1 << 14)                # the blocks of the smaller cases take the rebuild path too
        C, s, dp = np.full(n * d * d, np.nan), np.full(n * d, np.nan), np.full(n, np.nan)
        g.von_mises_host(PRM, d, n, deps, sigma_n, p, C, s, dp)
        np.testing.assert_array_equal(s, ref[1])
        np.testing.assert_array_equal(dp, ref[2])
        if rebuild and n < (1 << 18):                                 # the plain call above stayed in copy mode below its threshold
            assert_close_scaled(C, ref[0], 1e-14, "host-rebuilt tangent")
        else:
            np.testing.assert_array_equal(C, ref[0])
        with pytest.raises(ValueError, match="no communicator"):     # a local group has no exchange step
            g.all_gather([0] * n_dev, 16)
    finally:
        g.close()


def test_factory_over_several_local_pipelines(ctx, oracle):
    """make_von_mises(devices=[...]): the NumPy arrays of one call cut into one cell block per listed GPU (here device 0
    three times), results in the reference's tuple order, equal to the single-GPU factory."""
    n, nq, d = 200_000, 8, 6
    deps, sigma_n, p = vm_inputs(n, d, seed=14)
    one = make_von_mises(sigma_n, p, ctx=ctx, host_tangent="copy")
    many = make_von_mises(sigma_n, p, ctx=ctx, host_tangent="copy", devices=[0, 0, 0])
    a = one((1,))(deps.reshape(n // nq, nq, d))
    b = many((1,))(deps.reshape(n // nq, nq, d))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    Co, so, dpo = oracle.von_mises(deps, sigma_n, p)
    assert_close_scaled(b[0], Co.reshape(-1), 1e-13, "C_tang over three pipelines")
    with pytest.raises(NotImplementedError):
        many((0,))
    with pytest.raises(ValueError, match="devices"):
        make_von_mises(sigma_n, p, ctx=ctx, state="resident", devices=[0])
