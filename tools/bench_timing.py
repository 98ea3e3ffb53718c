"""The headline's timing protocol, the part that is plain arithmetic (warm-up, then >= 20 launches per batch, the
MEDIAN of 5 batches): which batch is reported, and what the line says about all of them. bench.py measures (fences, HIP events, the reduction
with MAX over ranks); this module only summarises, so that tests/test_bench_timing.py can check the rule on the CPU."""
from __future__ import annotations

import statistics


def median_batch(walls) -> int:
    """Index of the batch whose wall time is the median (the upper median for an even count): `value` and `ms_per_step` are that batch's."""
    return sorted(range(len(walls)), key=lambda b: walls[b])[len(walls) // 2]


def batch_record(walls, kernel_ms, gaps_us, steps: int, gather_on: bool = False):
    """(index of the median batch, the `timing` fields of the result line).

    walls      wall seconds of every batch of `steps` steps (under torch.distributed: already the maximum over ranks)
    kernel_ms  mean HIP-event time of the launches of every batch (this rank's)
    gaps_us    per batch, the intervals between the end of one step's launch and the start of the next one's (steps - 1 values)"""
    B = len(walls)
    b_med = median_batch(walls)
    gap_max = max(((g, b, k) for b, row in enumerate(gaps_us) for k, g in enumerate(row)), default=(0.0, 0, 0))
    flat = [g for row in gaps_us for g in row]
    return b_med, {
        "batches": B, "median_batch": b_med,
        "ms_per_step_batches": [t / steps * 1e3 for t in walls],
        "kernel_ms_batches": list(kernel_ms),
        "step_gap_us_max": gap_max[0], "step_gap_us_max_at": {"batch": gap_max[1], "after_step": gap_max[2]},
        "step_gap_us_median": statistics.median(flat) if flat else 0.0,
        "step_gap_meaning": ("interval between the end of one step's kernel and the start of the next step's, from the launch stream's own events"
                             + (" (with the gather on it contains the exchange)" if gather_on else "")),
    }
