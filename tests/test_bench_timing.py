"""The headline's timing rule on the CPU (not gpu): the median of the batches is what `value` / `ms_per_step` report, and a stalled batch or a
stalled step is visible on the line instead of being averaged in (one 16 ms batch had read 13 % low)."""
from tools.bench_timing import batch_record, median_batch


def test_a_stalled_batch_does_not_move_the_reported_step_time():
    K = 20
    walls = [0.01436, 0.01404, 0.01636, 0.01402, 0.01401]           # batch 2 lost 2.3 ms somewhere (the r05 driver run's pattern)
    kern = [0.702, 0.697, 0.698, 0.697, 0.696]
    gaps = [[6.0] * (K - 1) for _ in walls]
    gaps[2][7] = 2310.0                                             # ... between step 7 and step 8
    b, rec = batch_record(walls, kern, gaps, K)
    assert b == 1 and abs(walls[b] / K * 1e3 - 0.702) < 1e-9        # the median batch, not the mean (0.728 ms) nor the stalled one
    assert rec["batches"] == 5 and rec["median_batch"] == 1 and len(rec["ms_per_step_batches"]) == 5
    assert abs(rec["ms_per_step_batches"][2] - 0.818) < 1e-9        # the slow batch is on the line
    assert rec["step_gap_us_max"] == 2310.0 and rec["step_gap_us_max_at"] == {"batch": 2, "after_step": 7}
    assert rec["step_gap_us_median"] == 6.0 and "exchange" not in rec["step_gap_meaning"]


def test_median_rule_for_any_batch_count_and_one_step_batches():
    assert median_batch([3.0]) == 0
    assert median_batch([2.0, 1.0]) == 0                            # even count: the upper median
    assert median_batch([5.0, 1.0, 3.0]) == 2
    assert median_batch([4.0, 1.0, 3.0, 2.0]) == 2
    b, rec = batch_record([0.5, 0.4, 0.6], [1.0, 1.0, 1.0], [[], [], []], steps=1, gather_on=True)     # K = 1: no gaps to report
    assert b == 0 and rec["step_gap_us_max"] == 0.0 and rec["step_gap_us_median"] == 0.0 and "exchange" in rec["step_gap_meaning"]
