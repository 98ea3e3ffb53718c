"""Cell-block sharding + all-gather (not gpu): world_size-2 gloo run on CPU, plus partition arithmetic."""
import os
import socket
import sys

import numpy as np
import pytest

from dolfinx_external_operator_amd.sharding import CellBlockPartition


def test_partition_covers_every_cell_once():
    for num_cells, nq, world in [(125_000, 8, 8), (1001, 3, 2), (7, 3, 4), (0, 8, 2), (64, 1, 8), (10, 8, 16)]:
        part = CellBlockPartition(num_cells, nq, world)
        assert part.points_per_rank % 64 == 0
        seen = np.zeros(num_cells, dtype=int)
        for r in range(world):
            b, e = part.cell_range(r)
            assert 0 <= b <= e <= num_cells and e - b <= part.cells_per_rank
            seen[b:e] += 1
        assert np.all(seen == 1)
        assert part.padded_points >= part.num_points


def test_local_input_pads_with_zeros_and_trim_drops_them():
    part = CellBlockPartition(5, 8, 2)
    full = np.arange(5 * 8 * 6, dtype=float)
    a, b = part.local_input(full, 0, 6), part.local_input(full, 1, 6)
    assert a.size == b.size == part.points_per_rank * 6
    glued = np.concatenate([a, b])
    assert np.array_equal(part.trim(glued, 6), full)
    assert np.all(glued[full.size:] == 0)


def _worker(rank, world, port, num_cells, nq, d, ret):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dolfinx_external_operator_amd.sharding import CellBlockPartition, all_gather_flat, all_gather_flat_into
    from oracle import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(5))  # same global arrays on every rank
        n = num_cells * nq
        deps = rng.normal(0, 3e-3, (n, d))
        sigma_n = rng.normal(0, 100.0, (n, d))
        p = np.abs(rng.normal(0, 1e-3, n))
        part = CellBlockPartition(num_cells, nq, world)
        o = load_oracle()
        # this rank's padded block; the CPU oracle stands in for the HIP kernel (no GPU in this test)
        le = part.local_input(deps, rank, d).reshape(-1, d)
        ls = part.local_input(sigma_n, rank, d).reshape(-1, d)
        lp = part.local_input(p, rank, 1)
        with np.errstate(all="ignore"):
            C, s, dp = o.von_mises(le, ls, lp)
        gC = part.trim(all_gather_flat(torch.from_numpy(C.reshape(-1))), d * d).numpy()
        out = torch.empty(world * s.size, dtype=torch.float64)
        all_gather_flat_into(out, torch.from_numpy(s.reshape(-1)))
        gs = part.trim(out, d).numpy()
        gdp = part.trim(all_gather_flat(torch.from_numpy(dp)), 1).numpy()
        Cf, sf, dpf = o.von_mises(deps, sigma_n, p)
        ok = np.array_equal(gC, Cf.reshape(-1)) and np.array_equal(gs, sf.reshape(-1)) and np.array_equal(gdp, dpf)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_cells,nq,d", [(101, 8, 6), (50, 3, 4)])
def test_two_rank_gloo_gather_reassembles_the_flat_coefficient_vectors(oracle, num_cells, nq, d):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_cells, nq, d, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=180)
        assert pr.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def numpy_expand_tangent(sigma, dp, d, E=70e3, nu=0.3):
    """NumPy statement of the tangent-from-state formulas (dxo_vm_expand_tangent), the dp = -0.0 mark of the reference's
    0/0 point included: stand-in for the HIP rebuild kernel in the CPU-only gather test, itself checked against the
    oracle's tangent below."""
    lmbda, mu = E * nu / ((1 + nu) * (1 - 2 * nu)), E / (2 * (1 + nu))
    Et = E / 100.0
    H = E * Et / (E - Et)
    sigma = sigma.reshape(-1, d)
    one = np.zeros(d)
    one[:3] = 1.0
    C_el = lmbda * np.outer(one, one) + 2 * mu * np.eye(d)
    dev = np.eye(d) - np.outer(one, one) / 3.0
    s = sigma @ dev.T
    seq = np.sqrt(1.5 * np.sum(s * s, axis=1))
    with np.errstate(all="ignore"):
        beta = 3 * mu * dp / (seq + 3 * mu * dp)
        n = s / seq[:, None] * (dp > 0)[:, None]
    a = 3 * mu * (3 * mu / (3 * mu + H) - beta)
    a = np.where((dp == 0.0) & np.signbit(dp), np.nan, a)     # the producer's mark -> the reference's NaN tangent
    return C_el[None] - a[:, None, None] * n[:, :, None] * n[:, None, :] - (2 * mu * beta)[:, None, None] * dev[None]


def test_numpy_expand_matches_oracle_tangent(oracle):
    from conftest import assert_close_scaled, vm_inputs

    for d in (4, 6):
        deps, sigma_n, p = vm_inputs(3000, d, seed=11)
        deps[:1000] *= 0.2          # a third of the points stays elastic
        sigma_n[:1000] *= 0.2
        C, s, dp = oracle.von_mises(deps, sigma_n, p)
        assert 0.2 < (dp > 0).mean() < 0.95
        assert_close_scaled(numpy_expand_tangent(s, dp, d), C, 1e-13, f"tangent from state d={d}")


def _worker_compact(rank, world, port, num_cells, nq, d, ret, pipelined=0, identical=True):
    import hashlib

    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import vm_indeterminate_sigma0
    from dolfinx_external_operator_amd.sharding import (CellBlockPartition, gather_von_mises_compact, gather_von_mises_compact_direct,
                                                        gather_von_mises_compact_pipelined, remote_point_ranges)
    from oracle import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = load_oracle()
        rng = np.random.Generator(np.random.PCG64(6))
        n = num_cells * nq
        deps = rng.normal(0, 3e-3, (n, d))
        sigma_n = rng.normal(0, 100.0, (n, d))
        p = np.abs(rng.normal(0, 1e-3, n))
        # one point per rank block sits EXACTLY on the yield surface (f_elastic == 0): the reference's 0/0, NaN tangent (:318)
        sigma_0, sn_row = vm_indeterminate_sigma0(lambda e, s_, p_, s0: o.von_mises(e, s_, p_, sigma_0=s0), d)
        part = CellBlockPartition(num_cells, nq, world)
        m = part.points_per_rank
        special = [b + 3 for b, e in (part.point_range(r) for r in range(world)) if e - b > 3]
        for i in special:
            deps[i], sigma_n[i], p[i] = 0.0, sn_row, 0.0
        with np.errstate(all="ignore"):
            C, s, dp = o.von_mises(part.local_input(deps, rank, d).reshape(-1, d),
                                   part.local_input(sigma_n, rank, d).reshape(-1, d), part.local_input(p, rank, 1), sigma_0=sigma_0)
        # what the kernel does with option vm_mark_indeterminate = 1 (vm_core.h): dp = -0.0 where f_elastic == 0 exactly
        # (seen here as: tangent NaN, stress finite, dp == 0)
        marked = np.isnan(C.reshape(len(dp), -1)).all(axis=1) & np.isfinite(s.reshape(len(dp), -1)).all(axis=1) & (dp == 0.0)
        dp = np.where(marked, -0.0, dp)
        # the owner writes its block straight into the full-length buffers, as bench.py does on the GPU; with `identical`
        # its kernel writes no tangent at all (dxo_von_mises with C_tang = NULL)
        Cf = torch.full((world * m * d * d,), float("nan"), dtype=torch.float64)
        sf = torch.full((world * m * d,), float("nan"), dtype=torch.float64)
        dpf = torch.full((world * m,), float("nan"), dtype=torch.float64)
        if not identical:
            Cf[rank * m * d * d:(rank + 1) * m * d * d] = torch.from_numpy(C.reshape(-1))
        sf[rank * m * d:(rank + 1) * m * d] = torch.from_numpy(s.reshape(-1))
        dpf[rank * m:(rank + 1) * m] = torch.from_numpy(dp)
        calls = []

        def expand(sv, dv, Cv, npts):
            calls.append(npts)
            with np.errstate(all="ignore"):
                Cv.copy_(torch.from_numpy(numpy_expand_tangent(sv.numpy(), dv.numpy(), d).reshape(-1)))

        def clear_marks(dv, npts):
            a = dv.numpy()[:npts]
            a[(a == 0.0) & np.signbit(a)] = 0.0

        kw = dict(identical=identical, clear_marks=clear_marks)
        if pipelined < 0:       # the direct peer-to-peer form of the exchange
            gather_von_mises_compact_direct(Cf, sf, dpf, rank, d, expand, **kw)
        elif pipelined:
            gather_von_mises_compact_pipelined(Cf, sf, dpf, rank, d, expand, chunks=pipelined, **kw)
        else:
            gather_von_mises_compact(Cf, sf, dpf, rank, d, expand, **kw)
        with np.errstate(all="ignore"):
            Cw, sw, dpw = o.von_mises(deps, sigma_n, p, sigma_0=sigma_0)
        scale = np.nanmax(np.abs(Cw))
        gC = part.trim(Cf, d * d).numpy()
        gdp = part.trim(dpf, 1).numpy()
        own_b, own_e = part.point_range(rank)
        ok = np.array_equal(part.trim(sf, d).numpy(), sw.reshape(-1)) and np.array_equal(gdp, dpw)
        ok = ok and not np.signbit(gdp).any()                                      # marks cleared: the reference holds +0 there
        if not identical:   # the owner's block is the kernel's own tangent, untouched
            ok = ok and np.array_equal(gC[own_b * d * d:own_e * d * d], Cw.reshape(-1)[own_b * d * d:own_e * d * d], equal_nan=True)
        # NaN pattern of the whole gathered tangent == the reference's (the f_elastic == 0 points of EVERY block, remote ones included)
        nan_ref = np.isnan(Cw.reshape(-1))
        ok = ok and len(special) > 0 and all(np.isnan(Cw[i]).all() for i in special)
        ok = ok and np.array_equal(np.isnan(gC), nan_ref)
        ok = ok and np.max(np.abs(gC[~nan_ref] - Cw.reshape(-1)[~nan_ref])) <= 1e-13 * scale
        ok = ok and sum(calls) == (world if identical else world - 1) * m
        ok = ok and (pipelined > 0 or len(calls) == (1 if identical else len(remote_point_ranges(rank, world, m))))
        digest = hashlib.sha256(Cf.numpy().tobytes() + sf.numpy().tobytes() + dpf.numpy().tobytes()).hexdigest()
        ret[rank] = (bool(ok), digest)
    finally:
        dist.destroy_process_group()


def test_remote_point_ranges():
    from dolfinx_external_operator_amd.sharding import remote_point_ranges

    assert remote_point_ranges(0, 1, 64) == []
    assert remote_point_ranges(0, 4, 64) == [(64, 256)]
    assert remote_point_ranges(3, 4, 64) == [(0, 192)]
    assert remote_point_ranges(1, 4, 64) == [(0, 64), (128, 256)]
    with pytest.raises(ValueError):
        remote_point_ranges(4, 4, 64)


@pytest.mark.parametrize("num_cells,nq,d,world,pipelined,identical", [
    (101, 8, 6, 2, 0, True), (50, 3, 4, 3, 0, True), (101, 8, 6, 2, 3, True), (70, 8, 4, 3, 5, True),
    (101, 8, 6, 2, -1, True), (50, 3, 4, 3, -1, True), (64, 8, 6, 4, -1, True),
    (101, 8, 6, 2, 0, False), (70, 8, 4, 3, 5, False), (50, 3, 4, 3, -1, False)])
def test_gloo_compact_gather_rebuilds_tangents_nan_exact_and_replica_identical(oracle, num_cells, nq, d, world, pipelined, identical):
    """(sigma, dp) over the wire, tangents rebuilt locally: every rank must end with the reference's whole-range result —
    the NaN tangent at the f_elastic == 0 point of every block included (carried by the dp = -0.0 mark) — and, with
    `identical`, with the same BYTES on every rank."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_compact, args=(r, world, port, num_cells, nq, d, ret, pipelined, identical)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=180)
        assert pr.exitcode == 0
    got = dict(ret)
    assert {r: v[0] for r, v in got.items()} == {r: True for r in range(world)}
    if identical:
        assert len({v[1] for v in got.values()}) == 1, "replicas differ between ranks"


def _worker_in_place_real(rank, world, port, ret):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from dolfinx_external_operator_amd import sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DXO_GATHER_IN_PLACE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        real = sharding.all_gather_flat_into
        calls = {"aliased": 0, "cloned": 0}

        def counting(out, local, group=None, async_op=False):     # every rank REALLY issues the collective it was told to
            aliased = out.data_ptr() <= local.data_ptr() < out.data_ptr() + out.numel() * out.element_size()
            calls["aliased" if aliased else "cloned"] += 1
            return real(out, local, group, async_op)

        sharding.all_gather_flat_into = counting
        m = 96
        full = torch.zeros(world * m, dtype=torch.float64)
        forms = []
        for it, force in enumerate((None, True, True, False)):
            full.zero_()
            full[rank * m:(rank + 1) * m] = torch.arange(m, dtype=torch.float64) + 1000 * rank + it
            sharding.all_gather_in_place(full, rank, try_in_place=force)
            want = torch.cat([torch.arange(m, dtype=torch.float64) + 1000 * r + it for r in range(world)])
            assert torch.equal(full, want), (rank, it)
            forms.append(sharding.in_place_status()["ok"])
        # a raise out of the chosen form is fatal: it propagates, and NO second collective is issued in its place
        def refusing(out, local, group=None, async_op=False):
            calls["aliased"] += 1
            raise RuntimeError("backend refuses an input that aliases the output")

        sharding.all_gather_flat_into = refusing
        before = dict(calls)
        try:
            sharding.all_gather_in_place(full, rank, try_in_place=True)
            raised = False
        except RuntimeError:
            raised = True
        ret[rank] = (forms, calls["aliased"] - before["aliased"], calls["cloned"] - before["cloned"], raised, before)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_in_place_form_is_a_rule_all_ranks_share_and_a_raise_is_fatal(world):
    """The send-buffer form is decided from things every rank sees alike (argument > DXO_GATHER_IN_PLACE > backend) BEFORE any
    collective: gloo -> cloned by default; forced in-place, every rank really issues the aliased collective and the result is
    right; a raise propagates on the spot — no fallback collective that the other ranks would not match."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_in_place_real, args=(r, world, port, ret)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=180)
        assert pr.exitcode == 0
    for r in range(world):
        forms, aliased_after, cloned_after, raised, before = ret[r]
        assert forms == [False, True, True, False]
        assert before == {"aliased": 2, "cloned": 2}
        assert raised and aliased_after == 1 and cloned_after == 0


def test_in_place_rule_precedence(monkeypatch):
    from dolfinx_external_operator_amd.sharding import _in_place_rule

    monkeypatch.delenv("DXO_GATHER_IN_PLACE", raising=False)
    assert _in_place_rule("nccl", None)["ok"] is True and _in_place_rule("gloo", None)["ok"] is False
    monkeypatch.setenv("DXO_GATHER_IN_PLACE", "0")
    assert _in_place_rule("nccl", None)["ok"] is False and "DXO_GATHER_IN_PLACE=0" in _in_place_rule("nccl", None)["why"]
    assert _in_place_rule("nccl", True)["ok"] is True          # the argument wins
    monkeypatch.setenv("DXO_GATHER_IN_PLACE", "1")
    assert _in_place_rule("gloo", None)["ok"] is True and _in_place_rule("gloo", False)["ok"] is False


def test_chunk_backed_arena_tensors_are_refused_by_the_collectives(monkeypatch):
    """operators.make_von_mises documents that chunk-backed arena outputs must not go to RCCL / IPC and that sharding refuses
    them: the refusal, on the tagged tensor and on views of it."""
    import torch

    from dolfinx_external_operator_amd import sharding

    class Block:
        def __init__(self, kind):
            self.info = {"chosen_kind": kind}

    monkeypatch.delenv("DXO_ALLOW_VMM_COLLECTIVE", raising=False)
    t = torch.zeros(128, dtype=torch.float64)
    t.dxo_block = Block("2MB_chunks")
    for bad in (t, t[:64], t.view(2, 64)):
        with pytest.raises(ValueError, match="chunk-backed"):
            sharding.refuse_chunk_backed(bad)
    ok = torch.zeros(128, dtype=torch.float64)
    ok.dxo_block = Block("hipMalloc")
    sharding.refuse_chunk_backed(ok, ok[:64], torch.zeros(4))
    with pytest.raises(ValueError, match="chunk-backed"):
        sharding._check_full(torch.zeros(2 * 36), t[:2 * 6], torch.zeros(2), 1, 6)
    monkeypatch.setenv("DXO_ALLOW_VMM_COLLECTIVE", "1")
    sharding.refuse_chunk_backed(t)
