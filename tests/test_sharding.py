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
