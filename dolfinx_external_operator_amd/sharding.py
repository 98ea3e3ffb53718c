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


_INTO_TENSOR_OK: dict[str, bool] = {}


def _backend_has_into_tensor(group=None, device=None) -> bool:
    """Does the group's backend implement all_gather_into_tensor for tensors on `device`? Decided ONCE per
    (backend, device type) with a one-element
    collective (every rank runs it at its first gather, so the ranks stay in step); after that a RuntimeError
    out of a gather is a real failure and propagates instead of being retried through another code path."""
    import torch
    import torch.distributed as dist

    dev = torch.device(device) if device is not None else torch.device("cpu")
    backend = f"{dist.get_backend(group)}/{dev.type}"
    ok = _INTO_TENSOR_OK.get(backend)
    if ok is None:
        world = dist.get_world_size(group)
        probe_in = torch.zeros(1, dtype=torch.float64, device=dev)
        probe_out = torch.zeros(world, dtype=torch.float64, device=dev)
        try:
            dist.all_gather_into_tensor(probe_out, probe_in, group=group)
            ok = True
        except (RuntimeError, NotImplementedError):
            ok = False
        _INTO_TENSOR_OK[backend] = ok
    return ok


def all_gather_flat_into(out, local, group=None, async_op: bool = False):
    """All-gather into a caller-owned flat buffer (no allocation inside the timed region)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if out.numel() != world * local.numel():
        raise ValueError(f"gather buffer has {out.numel()} elements, need {world * local.numel()}")
    if _backend_has_into_tensor(group, local.device):
        return dist.all_gather_into_tensor(out, local, group=group, async_op=async_op)
    chunks = list(out.view(world, local.numel()).unbind(0))
    return dist.all_gather(chunks, local, group=group, async_op=async_op)


_IN_PLACE: dict = {}   # (backend, id(group)) -> {"ok": bool, "why": str}: a RULE every rank evaluates alike, never a trial


def in_place_status(group=None) -> dict | None:
    """The form `all_gather_in_place` uses on this group ({"ok": bool, "why": ...}), None before its first call."""
    import torch.distributed as dist

    return _IN_PLACE.get((dist.get_backend(group), id(group)))


def _in_place_rule(backend: str, try_in_place: bool | None) -> dict:
    """Which send buffer the gather uses — decided from things EVERY rank sees alike (argument, environment, backend name),
    before any rank issues a collective. There is no trial and no fallback: ranks that took different forms, or a rank that
    skipped a collective the others had already enqueued, would leave mismatched collective sequences on the communicator
    (a hang or silent corruption). A raise out of the chosen form is therefore fatal and propagates."""
    import os

    env = os.environ.get("DXO_GATHER_IN_PLACE", "")
    if try_in_place is not None:
        ok, why = bool(try_in_place), "try_in_place argument"
    elif env in ("0", "1"):
        ok, why = env == "1", f"DXO_GATHER_IN_PLACE={env}"
    else:
        ok, why = backend == "nccl", f"backend {backend}"
    return {"ok": ok, "why": why + (": in-place all_gather_into_tensor on the aliasing view (sendbuff == recvbuff + rank*count)" if ok
                                    else ": cloned send buffer")}


def refuse_chunk_backed(*tensors) -> None:
    """Arena blocks backed by 2 MB virtual-memory chunks (ctx option placement_vmm) cannot be exported with hipIpcGetMemHandle,
    which RCCL may use for peer access: handing one to a collective is undefined behaviour, so it is refused here. The arena
    tags its tensors (`dxo_block`); a view carries its base. DXO_ALLOW_VMM_COLLECTIVE=1 lifts the refusal (experiments)."""
    import os

    if os.environ.get("DXO_ALLOW_VMM_COLLECTIVE") == "1":
        return
    for t in tensors:
        for cand in (t, getattr(t, "_base", None)):
            blk = getattr(cand, "dxo_block", None) if cand is not None else None
            kind = (getattr(blk, "info", None) or {}).get("chosen_kind") if blk is not None else None
            if kind == "2MB_chunks":
                raise ValueError("this tensor lives in a chunk-backed (virtual-memory) block of the output arena, which must not be handed "
                                 "to RCCL / IPC: set ctx option placement_vmm = 0 before allocating outputs that go into a collective")


def all_gather_in_place(full, rank: int, group=None, *, try_in_place: bool | None = None):
    """All-gather where every rank has already written its block into `full` at [rank*m, (rank+1)*m).

    With RCCL the send buffer is that view itself (NCCL's documented in-place all-gather: sendbuff == recvbuff + rank*count —
    the form torch's own FSDP uses), so the kernel's output is never copied locally; other backends (gloo in the tests) get
    a clone of the block. The form is fixed by `_in_place_rule` before the first collective (argument > DXO_GATHER_IN_PLACE >
    backend name) and reported by `in_place_status`; a raise from the collective is fatal, not a reason to try another form."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if full.numel() % world:
        raise ValueError(f"gather buffer of {full.numel()} elements does not split into {world} equal blocks")
    refuse_chunk_backed(full)
    m = full.numel() // world
    local = full[rank * m:(rank + 1) * m]
    key = (dist.get_backend(group), id(group))
    st = _IN_PLACE.get(key)
    if st is None or try_in_place is not None:
        st = _IN_PLACE[key] = _in_place_rule(key[0], try_in_place)
    if not st["ok"]:
        local = local.clone()
    return all_gather_flat_into(full, local, group)


def remote_point_ranges(rank: int, world: int, points_per_rank: int) -> list[tuple[int, int]]:
    """[begin, end) point ranges of the blocks owned by OTHER ranks: at most two contiguous runs."""
    if not 0 <= rank < world:
        raise ValueError("rank outside [0, world)")
    runs = [(0, rank * points_per_rank), ((rank + 1) * points_per_rank, world * points_per_rank)]
    return [(b, e) for b, e in runs if e > b]


def _check_full(C_tang_full, sigma_full, dp_full, world: int, d: int) -> int:
    m = dp_full.numel() // world
    if dp_full.numel() != m * world or sigma_full.numel() != m * world * d or C_tang_full.numel() != m * world * d * d:
        raise ValueError("full buffers do not hold world equal blocks of (C_tang, sigma, dp)")
    refuse_chunk_backed(sigma_full, dp_full, C_tang_full)
    return m


def _rebuild_ranges(rank: int, world: int, m: int, identical: bool) -> list[tuple[int, int]]:
    return [(0, world * m)] if identical and m > 0 else remote_point_ranges(rank, world, m)


def gather_von_mises_compact(C_tang_full, sigma_full, dp_full, rank: int, d: int, expand, group=None, *, identical: bool = False,
                             clear_marks=None) -> None:
    """Reassemble (C_tang, sigma, dp) on every rank while moving only (sigma, dp) over the links.

    xGMI, not HBM, bounds the reassembly: the full outputs are (d*d+d+1) doubles per point (344 B at d = 6), of
    which the tangent is d*d. The tangent is a function of the returned state (see dxo_vm_expand_tangent), so
    the ranks all-gather sigma and dp in place ((d+1) doubles, 56 B at d = 6: 6.1x fewer link bytes) and rebuild
    the tangents locally — read 56 B + write 288 B per point at HBM speed, ~5 ms for 8*10^7 points, against ~50 ms
    saved on the links at 8 GPUs.

    `identical=False` (default): only the remote blocks are rebuilt and the owner keeps the tangent its kernel wrote,
    including the reference's NaN tangent at f_el == 0 (the replicas then agree to rounding, <= 1e-14 of the scale, not bit
    for bit). `identical=True` (what bench.py uses): EVERY block's tangent is rebuilt, the rank's own included, in one
    launch over the whole range — all ranks run the same arithmetic on the same gathered values, so the replicas of the
    coefficient vector are bit-identical across ranks; the owner's kernel then need not write a tangent at all (dxo_von_mises
    with C_tang = NULL), but it MUST have run with option "vm_mark_indeterminate" and `clear_marks` must be given, or the
    owner's NaN tangents are rebuilt as C_elas.

    The reference's 0/0 point (f_elastic == 0 exactly, demo_plasticity_von_mises.py:318: NaN tangent) leaves no trace in
    the VALUES of (sigma, dp); a producer run with option "vm_mark_indeterminate" returns dp = -0.0 there, the rebuild
    turns the mark into the reference's NaN tangent, and `clear_marks(dp_view, n_points)` (Context.vm_clear_marks) —
    called on the whole dp array at the end when given — restores the reference's +0.

    `expand(sigma_view, dp_view, C_tang_view, n_points)` launches the rebuild for one contiguous run."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    m = _check_full(C_tang_full, sigma_full, dp_full, world, d)
    all_gather_in_place(sigma_full, rank, group)
    all_gather_in_place(dp_full, rank, group)
    for b, e in _rebuild_ranges(rank, world, m, identical):
        expand(sigma_full[b * d:e * d], dp_full[b:e], C_tang_full[b * d * d:e * d * d], e - b)
    if clear_marks is not None and m > 0:
        clear_marks(dp_full, world * m)


def gather_von_mises_compact_pipelined(C_tang_full, sigma_full, dp_full, rank: int, d: int, expand, chunks: int = 4,
                                       group=None, *, identical: bool = False, clear_marks=None) -> None:
    """`gather_von_mises_compact` with the tangent rebuild overlapped with the link traffic (SURVEY.md 8e iii).

    Every rank's block of m points is cut into `chunks` pieces on 64-point borders. All pieces are put on the wire
    at once as asynchronous all-gathers (they queue on the collective stream in order); as soon as piece k of
    (sigma, dp) has arrived from every rank, the tangents of piece k are rebuilt on the compute stream while
    pieces k+1.. are still in flight. The rebuild (~5 ms per step at 8 GPUs) disappears behind the gather; the cost
    is `chunks` smaller collectives instead of one and, with the list form of all_gather, a staging copy inside the
    backend. Same result as the unpipelined form, bit for bit (`identical`, `clear_marks`: as there)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    m = _check_full(C_tang_full, sigma_full, dp_full, world, d)
    if chunks < 1:
        raise ValueError("chunks >= 1 required")
    step = -(-m // chunks)
    step = -(-step // WAVE_TILE) * WAVE_TILE
    pieces = [(b, min(b + step, m)) for b in range(0, m, step)]
    works = []
    for b, e in pieces:
        s_out = [sigma_full[(r * m + b) * d:(r * m + e) * d] for r in range(world)]
        p_out = [dp_full[r * m + b:r * m + e] for r in range(world)]
        # the send views are cloned: the list form may stage its outputs, and an output view must not alias the input
        ws = dist.all_gather(s_out, s_out[rank].clone(), group=group, async_op=True)
        wp = dist.all_gather(p_out, p_out[rank].clone(), group=group, async_op=True)
        works.append((ws, wp))
    for (b, e), (ws, wp) in zip(pieces, works):
        ws.wait()
        wp.wait()
        for r in range(world):
            if r == rank and not identical:
                continue
            lo, hi = r * m + b, r * m + e
            expand(sigma_full[lo * d:hi * d], dp_full[lo:hi], C_tang_full[lo * d * d:hi * d * d], hi - lo)
    if clear_marks is not None and m > 0:
        clear_marks(dp_full, world * m)


def exchange_blocks_direct(full, rank: int, group=None) -> None:
    """The same result as `all_gather_in_place`, as point-to-point traffic: every rank sends its block to every peer
    and receives each peer's block straight into place, all 2 (world - 1) operations in ONE batch (ncclGroupStart / End
    with RCCL). On a fully connected xGMI node (one link per GPU pair) that puts each block on its own link at once —
    the all-pairs pattern SURVEY.md 8e asks for — whatever algorithm the library's all-gather would have chosen for the
    message size (a ring moves world - 1 blocks over every link, one after another)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if full.numel() % world:
        raise ValueError(f"gather buffer of {full.numel()} elements does not split into {world} equal blocks")
    if world == 1:
        return
    refuse_chunk_backed(full)
    m = full.numel() // world
    own = full[rank * m:(rank + 1) * m]
    ops = []
    for step in range(1, world):          # peer order staggered by rank: at every step the pairs are disjoint
        to, frm = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, own, to, group))
        ops.append(dist.P2POp(dist.irecv, full[frm * m:(frm + 1) * m], frm, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def gather_von_mises_compact_direct(C_tang_full, sigma_full, dp_full, rank: int, d: int, expand, group=None, *,
                                    identical: bool = False, clear_marks=None) -> None:
    """`gather_von_mises_compact` with the exchange of (sigma, dp) as direct peer-to-peer sends / receives
    (`exchange_blocks_direct`) instead of the backend's all-gather; the rebuild of the tangents is the same."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    m = _check_full(C_tang_full, sigma_full, dp_full, world, d)
    exchange_blocks_direct(sigma_full, rank, group)
    exchange_blocks_direct(dp_full, rank, group)
    for b, e in _rebuild_ranges(rank, world, m, identical):
        expand(sigma_full[b * d:e * d], dp_full[b:e], C_tang_full[b * d * d:e * d * d], e - b)
    if clear_marks is not None and m > 0:
        clear_marks(dp_full, world * m)


__all__ = ["CellBlockPartition", "in_place_status", "refuse_chunk_backed", "exchange_blocks_direct", "gather_von_mises_compact_direct", "all_gather_flat", "all_gather_flat_into", "all_gather_in_place",
           "remote_point_ranges", "gather_von_mises_compact", "gather_von_mises_compact_pipelined", "WAVE_TILE"]
