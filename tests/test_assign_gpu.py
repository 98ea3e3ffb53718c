"""Device assigners (dxo_assign) against the value-side mirrors of the reference's assigners in evaluation.py
(_assign_non_mixed external_operator.py:286-287, _assign_mixed_2d :292-311, _assign_mixed_3d :313-335) — bit for bit,
including NumPy's last-writer-wins rule where a continuous space shares dofs between cells."""
import numpy as np
import pytest

from dolfinx_external_operator_amd import (AssignDesc, MixedExternalOperator, QuadratureExternalOperator,
                                           get_unrolled_dofmap)

pytestmark = pytest.mark.gpu


DTYPES = [np.float32, np.float64, np.complex128]      # the scalar types the reference's own tests run (test/test_multiaction.py:15-23)


def as_dtype(rng, shape, dtype):
    """Seeded values of `dtype` (complex: independent real and imaginary parts)."""
    v = rng.normal(size=shape)
    if np.issubdtype(dtype, np.complexfloating):
        v = v + 1j * rng.normal(size=shape)
    return v.astype(dtype)


def device_assign(ctx, desc, flat_dofs, values, coeff_size, initial):
    """dxo_assign, and the same assignment through a plan (dxo_assign_plan_create + dxo_assign_apply, applied twice: a plan
    is made once per dofmap and reused): both must leave the same bits. The element width travels in desc.elem_bytes and is
    taken from `values`' dtype here (float32 / float64 / complex128: the device moves 4 / 8 / 16-byte words)."""
    import torch

    values = np.ascontiguousarray(values).reshape(-1)
    assert initial.dtype == values.dtype
    desc.elem_bytes = values.dtype.itemsize
    d = torch.from_numpy(np.ascontiguousarray(flat_dofs, dtype=np.int32)).cuda()
    v = torch.from_numpy(values).cuda()
    c = torch.from_numpy(initial.copy()).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.assign(desc, d.data_ptr(), v.data_ptr(), c.data_ptr(), coeff_size)
    torch.cuda.synchronize()
    # the last-writer pass keeps 32-bit owner words at these sizes; the 64-bit words that > 2^32 - 2 entries take must give the same
    # array and the same plan (option assign_owner_bits = 64 forces them)
    ctx.set_option("assign_owner_bits", 64)
    try:
        c_wide = torch.from_numpy(initial.copy()).cuda()
        ctx.assign(desc, d.data_ptr(), v.data_ptr(), c_wide.data_ptr(), coeff_size)
        plan_wide = ctx.assign_plan(desc, d.data_ptr(), coeff_size)
        c_wide_plan = torch.from_numpy(initial.copy()).cuda()
        plan_wide.apply(v.data_ptr(), c_wide_plan.data_ptr())
        torch.cuda.synchronize()
        plan_wide.close()
    finally:
        ctx.set_option("assign_owner_bits", 0)
    as_real = lambda t: torch.view_as_real(t) if t.is_complex() else t
    assert torch.equal(as_real(c_wide), as_real(c)) and torch.equal(as_real(c_wide_plan), as_real(c))
    # the plan in SOURCE order (what a large plan may choose at its first apply; assign_plan_form = 2 builds it at any size)
    ctx.set_option("assign_plan_form", 2)
    try:
        plan_src = ctx.assign_plan(desc, d.data_ptr(), coeff_size)
    finally:
        ctx.set_option("assign_plan_form", 0)
    assert plan_src.form()["form"] == (2 if np.asarray(flat_dofs).size else 1)
    c_src = torch.from_numpy(initial.copy()).cuda()
    plan_src.apply(v.data_ptr(), c_src.data_ptr())
    torch.cuda.synchronize()
    plan_src.close()
    assert torch.equal(as_real(c_src), as_real(c))
    plan = ctx.assign_plan(desc, d.data_ptr(), coeff_size)
    del d                                                   # the plan does not keep the dofmap
    for _ in range(2):
        c2 = torch.from_numpy(initial.copy()).cuda()
        plan.apply(v.data_ptr(), c2.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(c2) if c2.is_complex() else c2, torch.view_as_real(c) if c.is_complex() else c)
    plan.close()
    return c.cpu().numpy()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bs", [1, 2, 3])
def test_non_mixed_unrolled_dofmap_with_shared_dofs(ctx, bs, dtype):
    """A P1-like continuous space on a strip of cells: neighbouring cells share nodes, values differ per cell, so
    the result depends on the write order — it must be NumPy's."""
    rng = np.random.Generator(np.random.PCG64(bs))
    n_cells, n_pts = 5000, 4
    dofmap = np.stack([np.arange(n_cells) + k for k in range(n_pts)], axis=1).astype(np.int32)   # heavy sharing
    rng.shuffle(dofmap, axis=0)
    unrolled = get_unrolled_dofmap(dofmap, bs)
    size = (n_cells + n_pts) * bs
    op = QuadratureExternalOperator(num_cells=n_cells, num_points=n_pts, value_shape=(bs,) if bs > 1 else (),
                                    unrolled_dofmap=unrolled, coefficient_size=size, dtype=dtype)
    values = as_dtype(rng, n_cells * n_pts * bs, dtype)
    op.ref_coefficient.x.array[:] = -3.0
    op._assign_func(values)
    assert op.ref_coefficient.x.array.dtype == dtype
    desc = AssignDesc(n_cells, n_pts, bs, 0, n_pts, bs, 0)
    got = device_assign(ctx, desc, unrolled, values, size, np.full(size, -3.0, dtype=dtype))
    assert got.dtype == dtype and np.array_equal(got, op.ref_coefficient.x.array)


@pytest.mark.parametrize("dtype", DTYPES)
def test_mixed_scalar_and_padded_vector_subspaces(ctx, dtype):
    rng = np.random.Generator(np.random.PCG64(11))
    n_cells = 1200
    # subspace 0: vector (2 components, 3 points), subspace 1: scalar (4 points); comp_size = 2 (padded), :151-161
    dm0 = rng.permutation(n_cells * 6).reshape(n_cells, 6).astype(np.int32)
    dm1 = (n_cells * 6 + rng.integers(0, n_cells, size=(n_cells, 4))).astype(np.int32)        # shared scalar dofs
    size = n_cells * 7
    sub = [{"n_pts": 3, "val_size": 2, "dofmap": dm0}, {"n_pts": 4, "val_size": 1, "dofmap": dm1}]
    op = MixedExternalOperator(num_cells=n_cells, subspaces=sub, coefficient_size=size, dtype=dtype)
    values = as_dtype(rng, n_cells * 7 * 2, dtype)
    op.ref_coefficient.x.array[:] = 9.0
    op._assign_func(values)                                                                    # _assign_mixed_3d
    coeff = np.full(size, 9.0, dtype=dtype)
    for info in op._mixed_subspace_info:
        desc = AssignDesc(n_cells, info["n_pts"], info["val_size"], info["offset"], op._n_points_total, op._comp_size, 0)
        coeff = device_assign(ctx, desc, info["flat_dofs"], values, size, coeff)
    assert np.array_equal(coeff, op.ref_coefficient.x.array)
    # all-scalar mixed space -> _assign_mixed_2d
    sub2 = [{"n_pts": 3, "val_size": 1, "dofmap": dm0[:, :3]}, {"n_pts": 4, "val_size": 1, "dofmap": dm1}]
    op2 = MixedExternalOperator(num_cells=n_cells, subspaces=sub2, coefficient_size=size, dtype=dtype)
    v2 = as_dtype(rng, n_cells * 7, dtype)
    op2._assign_func(v2)
    coeff = np.zeros(size, dtype=dtype)
    for info in op2._mixed_subspace_info:
        desc = AssignDesc(n_cells, info["n_pts"], 1, info["offset"], op2._n_points_total, 1, 0)
        coeff = device_assign(ctx, desc, info["flat_dofs"], v2, size, coeff)
    assert np.array_equal(coeff, op2.ref_coefficient.x.array)


@pytest.mark.parametrize("dtype", DTYPES)
def test_device_assigner_is_the_operators_assign_func_on_device_arrays(ctx, dtype):
    """DeviceAssigner(op, ctx).apply(values, coeff) against op._assign_func(values) for the four assigners the reference's constructor chooses
    from (external_operator.py:195-209): contiguous, unrolled dofmap (block size 2, shared dofs), mixed all-scalar (2-D), mixed padded (3-D)."""
    import torch

    from dolfinx_external_operator_amd import DeviceAssigner

    rng = np.random.Generator(np.random.PCG64(21))
    n_cells = 900
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dm = np.stack([np.arange(n_cells) + k for k in range(3)], axis=1).astype(np.int32)
    rng.shuffle(dm, axis=0)
    dm0 = rng.permutation(n_cells * 6).reshape(n_cells, 6).astype(np.int32)
    dm1 = (n_cells * 6 + rng.integers(0, n_cells, size=(n_cells, 4))).astype(np.int32)
    ops = {
        "contiguous": QuadratureExternalOperator(num_cells=n_cells, num_points=3, value_shape=(2, 2), dtype=dtype),
        "non_mixed": QuadratureExternalOperator(num_cells=n_cells, num_points=3, value_shape=(2,), unrolled_dofmap=get_unrolled_dofmap(dm, 2),
                                                coefficient_size=(n_cells + 3) * 2, dtype=dtype),
        "mixed_2d": MixedExternalOperator(num_cells=n_cells, subspaces=[{"n_pts": 3, "val_size": 1, "dofmap": dm0[:, :3]}, {"n_pts": 4, "val_size": 1, "dofmap": dm1}],
                                          coefficient_size=n_cells * 7, dtype=dtype),
        "mixed_3d": MixedExternalOperator(num_cells=n_cells, subspaces=[{"n_pts": 3, "val_size": 2, "dofmap": dm0}, {"n_pts": 4, "val_size": 1, "dofmap": dm1}],
                                          coefficient_size=n_cells * 7, dtype=dtype),
    }
    for kind, op in ops.items():
        da = DeviceAssigner(op, ctx)
        assert da.kind == kind and da.dtype == dtype
        values = as_dtype(rng, da.values_size, dtype)
        op.ref_coefficient.x.array[:] = 4.0
        op._assign_func(values)
        v = torch.from_numpy(values).cuda()
        c = torch.from_numpy(np.full(da.coeff_size, 4.0, dtype=dtype)).cuda()
        da.apply(v.data_ptr(), c.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), op.ref_coefficient.x.array), kind
        if kind == "contiguous":
            da.apply(c.data_ptr(), c.data_ptr())             # the operator wrote in place: nothing to move
        else:
            assert all(f["form"] == 1 for f in da.forms())    # small plans keep the entry-order form
        da.close()
    with pytest.raises(TypeError):
        DeviceAssigner(QuadratureExternalOperator(num_cells=4, num_points=2, dtype=np.float16), ctx)


def test_large_plan_times_both_orders_at_its_first_apply_and_keeps_one(ctx):
    """A plan of 2^20 coefficient entries or more carries the assignment by coefficient entry AND by position in `values`; its first apply
    launches both on the caller's arrays, times them and keeps the faster (include/dxo.h: dxo_assign_plan_form). Whatever it keeps, the
    coefficient is NumPy's `coeff[dofs] = values` (external_operator.py:286-287), on a Q2-hexahedra dofmap whose neighbouring cells share nodes."""
    import torch

    from tools.synthetic import structured_mesh_cached

    m = structured_mesh_cached("hexahedron", (52, 52, 52), 2, distort=0.0, seed=0)
    nc, npt = m.dofmap.shape
    size = m.node_x.shape[0]
    assert size >= 1 << 20
    rng = np.random.Generator(np.random.PCG64(11))
    values = rng.normal(size=nc * npt)
    expect = np.full(size, -7.0)
    expect[m.dofmap.reshape(-1)] = values                      # NumPy: the last writer wins
    d = torch.from_numpy(np.ascontiguousarray(m.dofmap.reshape(-1), dtype=np.int32)).cuda()
    v = torch.from_numpy(values).cuda()
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    desc = AssignDesc(nc, npt, 1, 0, npt, 1, 8)
    plan = ctx.assign_plan(desc, d.data_ptr(), size)
    assert plan.form()["form"] == 0                            # both orders on board, none chosen yet
    forms = []
    for _ in range(3):
        c = torch.full((size,), -7.0, dtype=torch.float64, device="cuda")
        plan.apply(v.data_ptr(), c.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), expect)
        forms.append(plan.form())
    assert forms[0]["form"] in (1, 2) and forms[0] == forms[1] == forms[2]      # decided once
    assert forms[0]["ms_dof_order"] > 0.0 and forms[0]["ms_source_order"] > 0.0
    fast = min(forms[0]["ms_dof_order"], forms[0]["ms_source_order"])
    assert (forms[0]["ms_source_order"] == fast) == (forms[0]["form"] == 2)
    plan.close()
    # either order on its own gives the same array
    for f in (1, 2):
        ctx.set_option("assign_plan_form", f)
        try:
            pf = ctx.assign_plan(desc, d.data_ptr(), size)
        finally:
            ctx.set_option("assign_plan_form", 0)
        assert pf.form()["form"] == f
        c = torch.full((size,), -7.0, dtype=torch.float64, device="cuda")
        pf.apply(v.data_ptr(), c.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(c.cpu().numpy(), expect)
        pf.close()
    with pytest.raises(ValueError):
        ctx.set_option("assign_plan_form", 3)


def test_large_plan_applied_inside_a_graph_capture_decides_later(ctx):
    """The first apply of a two-order plan waits on events, which a stream under capture cannot do: a captured apply takes the entry-order form
    and leaves the plan undecided; the replayed graph assigns like NumPy, and the next ordinary apply makes the choice."""
    import torch

    from tools.synthetic import structured_mesh_cached

    m = structured_mesh_cached("hexahedron", (52, 52, 52), 2, distort=0.0, seed=0)
    nc, npt = m.dofmap.shape
    size = m.node_x.shape[0]
    rng = np.random.Generator(np.random.PCG64(12))
    values = rng.normal(size=nc * npt)
    expect = np.full(size, 2.5)
    expect[m.dofmap.reshape(-1)] = values
    d = torch.from_numpy(np.ascontiguousarray(m.dofmap.reshape(-1), dtype=np.int32)).cuda()
    v = torch.from_numpy(values).cuda()
    c = torch.full((size,), 2.5, dtype=torch.float64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    plan = ctx.assign_plan(AssignDesc(nc, npt, 1, 0, npt, 1, 8), d.data_ptr(), size)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)      # the capture stream
            plan.apply(v.data_ptr(), c.data_ptr())
    finally:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    assert plan.form()["form"] == 0
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), expect)
    c.fill_(2.5)
    plan.apply(v.data_ptr(), c.data_ptr())
    torch.cuda.synchronize()
    assert plan.form()["form"] in (1, 2) and np.array_equal(c.cpu().numpy(), expect)
    plan.close()


def test_argument_checks_and_empty(ctx):
    import torch

    t = torch.zeros(8, dtype=torch.float64, device="cuda")
    i = torch.zeros(8, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ctx.assign(AssignDesc(2, 2, 2, 0, 2, 1, 0), i.data_ptr(), t.data_ptr(), t.data_ptr(), 8)   # comp_size < val_size
    with pytest.raises(ValueError):
        ctx.assign(AssignDesc(2, 2, 1, 3, 4, 1, 0), i.data_ptr(), t.data_ptr(), t.data_ptr(), 8)   # offset + n_pts > total
    ctx.assign(AssignDesc(0, 2, 1, 0, 2, 1, 0), None, None, None, 0)
    with pytest.raises(ValueError):
        ctx.assign_plan(AssignDesc(2, 2, 2, 0, 2, 1, 0), i.data_ptr(), 8)                          # comp_size < val_size
    bad = torch.tensor([0, 1, 99, 2], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="outside"):
        ctx.assign_plan(AssignDesc(2, 2, 1, 0, 2, 1, 0), bad.data_ptr(), 8)                        # NumPy raises IndexError here
    empty = ctx.assign_plan(AssignDesc(0, 2, 1, 0, 2, 1, 0), None, 0)
    empty.apply(None, None)
    for eb in (1, 2, 3, 12, 32):                                                                   # element widths: 4, 8, 16 (0 = 8)
        with pytest.raises(ValueError, match="elem_bytes"):
            ctx.assign(AssignDesc(2, 2, 1, 0, 2, 1, eb), i.data_ptr(), t.data_ptr(), t.data_ptr(), 8)
        with pytest.raises(ValueError, match="elem_bytes"):
            ctx.assign_plan(AssignDesc(2, 2, 1, 0, 2, 1, eb), i.data_ptr(), 8)
    with pytest.raises(ValueError, match="aligned"):                                               # complex128 moves 16-byte words
        ctx.assign(AssignDesc(2, 2, 1, 0, 2, 1, 16), i.data_ptr(), t.data_ptr() + 8, t.data_ptr(), 4)
