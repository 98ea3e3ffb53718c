"""Cell-block sharding of the quadrature-point range over the GPUs of one node, and the all-gather
that reassembles the flat coefficient arrays (BASELINE north_star).

The reference never gathers quadrature data: each MPI rank evaluates its own mesh partition
(src/dolfinx_external_operator/external_operator.py:365-371) and only halo-updates the coefficient
(:445). The design here is the north-star's: ONE logical coefficient vector, cells split into
contiguous blocks (arrays are cell-major, so a block of cells is a contiguous slice of every input and
output array), one process per GPU computes its block, and an all-gather over RCCL/xGMI gives every
rank the whole vector.

No data-path collective is needed for the kernels themselves (pointwise maps); the all-gather is the
only exchange step. `torch.distributed` is used as plumbing: backend "nccl" is RCCL on ROCm, "gloo" on
CPU for the tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

WAVE_TILE = 64  # points per wave tile of the HIP kernels; shard borders stay on tile borders


@dataclass(frozen=True)
class CellBlockPartition:
    """Split `num_cells` cells (each `nq` points) into `world` contiguous, equally sized blocks.

    Every rank owns `cells_per_rank` cells; the block size is rounded up so that
    `cells_per_rank * nq` is a multiple of 64 points (whole wave tiles, 16-byte aligned slices) and the
    last blocks may be partly or wholly padding. Equal blocks make the gather a plain all-gather
    (no all-gatherv): the padded tail is computed on zeros and dropped by `trim`.
    """
    num_cells: int
    nq: int
    world: int

    def __post_init__(self):
        if self.num_cells < 0 or self.nq <= 0 or self.world <= 0:
            raise ValueError("num_cells >= 0, nq > 0, world > 0 required")

    @property
    def cells_per_rank(self) -> int:
        per = -(-self.num_cells // self.world)  # ceil
        step = WAVE_TILE // np.gcd(WAVE_TILE, self.nq)  # smallest cell count with a whole number of tiles
        return int(-(-per // step) * step) if per > 0 else 0

    @property
    def points_per_rank(self) -> int:
        return self.cells_per_rank * self.nq

    @property
    def padded_points(self) -> int:
        return self.points_per_rank * self.world

    @property
    def num_points(self) -> int:
        return self.num_cells * self.nq

    def cell_range(self, rank: int) -> tuple[int, int]:
        """[begin, end) of the REAL cells of `rank` (may be empty for trailing ranks)."""
        b = min(rank * self.cells_per_rank, self.num_cells)
        e = min((rank + 1) * self.cells_per_rank, self.num_cells)
        return b, e

    def point_range(self, rank: int) -> tuple[int, int]:
        b, e = self.cell_range(rank)
        return b * self.nq, e * self.nq

    def local_input(self, full: np.ndarray, rank: int, width: int) -> np.ndarray:
        """This rank's padded slice of a flat per-point array with `width` values per point."""
        out = np.zeros(self.points_per_rank * width, dtype=full.dtype)
        b, e = self.point_range(rank)
        out[: (e - b) * width] = np.asarray(full).reshape(-1)[b * width: e * width]
        return out

    def trim(self, gathered, width: int):
        """Drop the padding of a gathered flat array: rank blocks are contiguous, so only the tail is padding."""
        return gathered[: self.num_points * width]


def all_gather_flat(local, group=None):
    """All-gather equally sized flat shards into one flat tensor ordered by rank (rank-major = cell-major).

    `local` is a 1-D torch tensor (CUDA for RCCL, CPU for gloo). Returns a new tensor of
    world * local.numel() elements. Uses `all_gather_into_tensor` (one RCCL all-gather, no per-rank list
    concatenation); falls back to `all_gather` on backends that lack it.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    all_gather_flat_into(out, local, group)
    return out


def all_gather_flat_into(out, local, group=None, async_op: bool = False):
    """All-gather into a caller-owned flat buffer (no allocation inside the timed region)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out.numel() != world * local.numel():
        raise ValueError(f"gather buffer has {out.numel()} elements, need {world * local.numel()}")
    try:
        return dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):
        if async_op:
            raise
        chunks = list(out.view(world, local.numel()).unbind(0))
        return dist.all_gather(chunks, local, group=group)


__all__ = ["CellBlockPartition", "all_gather_flat", "all_gather_flat_into", "WAVE_TILE"]
