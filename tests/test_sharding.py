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
    """NumPy statement of the tangent-from-state formulas (dxo_vm_expand_tangent): stand-in for the HIP rebuild
    kernel in the CPU-only gather test, itself checked against the oracle's tangent below."""
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


def _worker_compact(rank, world, port, num_cells, nq, d, ret, pipelined=0):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dolfinx_external_operator_amd.sharding import (CellBlockPartition, gather_von_mises_compact, gather_von_mises_compact_direct,
                                                        gather_von_mises_compact_pipelined, remote_point_ranges)
    from oracle import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(6))
        n = num_cells * nq
        deps = rng.normal(0, 3e-3, (n, d))
        sigma_n = rng.normal(0, 100.0, (n, d))
        p = np.abs(rng.normal(0, 1e-3, n))
        part = CellBlockPartition(num_cells, nq, world)
        m = part.points_per_rank
        o = load_oracle()
        with np.errstate(all="ignore"):
            C, s, dp = o.von_mises(part.local_input(deps, rank, d).reshape(-1, d),
                                   part.local_input(sigma_n, rank, d).reshape(-1, d), part.local_input(p, rank, 1))
        # the owner writes its block straight into the full-length buffers, as bench.py does on the GPU
        Cf = torch.full((world * m * d * d,), float("nan"), dtype=torch.float64)
        sf = torch.full((world * m * d,), float("nan"), dtype=torch.float64)
        dpf = torch.full((world * m,), float("nan"), dtype=torch.float64)
        Cf[rank * m * d * d:(rank + 1) * m * d * d] = torch.from_numpy(C.reshape(-1))
        sf[rank * m * d:(rank + 1) * m * d] = torch.from_numpy(s.reshape(-1))
        dpf[rank * m:(rank + 1) * m] = torch.from_numpy(dp)
        calls = []

        def expand(sv, dv, Cv, npts):
            calls.append(npts)
            with np.errstate(all="ignore"):
                Cv.copy_(torch.from_numpy(numpy_expand_tangent(sv.numpy(), dv.numpy(), d).reshape(-1)))

        if pipelined < 0:       # the direct peer-to-peer form of the exchange
            gather_von_mises_compact_direct(Cf, sf, dpf, rank, d, expand)
        elif pipelined:
            gather_von_mises_compact_pipelined(Cf, sf, dpf, rank, d, expand, chunks=pipelined)
        else:
            gather_von_mises_compact(Cf, sf, dpf, rank, d, expand)
        with np.errstate(all="ignore"):
            Cw, sw, dpw = o.von_mises(deps, sigma_n, p)
        scale = np.abs(Cw).max()
        gC = part.trim(Cf, d * d).numpy()
        own_b, own_e = part.point_range(rank)
        ok = np.array_equal(part.trim(sf, d).numpy(), sw.reshape(-1)) and np.array_equal(part.trim(dpf, 1).numpy(), dpw)
        ok = ok and np.array_equal(gC[own_b * d * d:own_e * d * d], Cw.reshape(-1)[own_b * d * d:own_e * d * d])  # untouched
        ok = ok and np.max(np.abs(gC - Cw.reshape(-1))) <= 1e-13 * scale
        ok = ok and sum(calls) == (world - 1) * m
        ok = ok and (pipelined > 0 or len(calls) == len(remote_point_ranges(rank, world, m)))
        ret[rank] = bool(ok)
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


@pytest.mark.parametrize("num_cells,nq,d,world,pipelined", [(101, 8, 6, 2, 0), (50, 3, 4, 3, 0), (101, 8, 6, 2, 3), (70, 8, 4, 3, 5),
                                                            (101, 8, 6, 2, -1), (50, 3, 4, 3, -1), (64, 8, 6, 4, -1)])
def test_gloo_compact_gather_rebuilds_remote_tangents(oracle, num_cells, nq, d, world, pipelined):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_compact, args=(r, world, port, num_cells, nq, d, ret, pipelined)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=180)
        assert pr.exitcode == 0
    assert dict(ret) == {r: True for r in range(world)}
