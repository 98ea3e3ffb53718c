"""The stdout line of bench.py stays inside the driver's reach (not gpu).

Round 4's line was the whole 38 KB record and `BENCH_r04.json.parsed` came back null. tools/bench_line.py now writes a bounded
extract; these tests build it from canned records — round 4's real one (tests/golden/bench_full_r04.json) and an inflated one —
and read it back the way a driver with an 8 KB tail would.
"""
import copy
import json
import pathlib

from tools.bench_line import LINE_BUDGET, compact_line, compact_record

ROOT = pathlib.Path(__file__).resolve().parents[1]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def canned():
    return json.loads((ROOT / "tests" / "golden" / "bench_full_r04.json").read_text())


def test_round4_record_fits_and_parses_from_an_8k_tail():
    full = canned()
    assert len(json.dumps(full)) > 30_000      # the record that was lost
    stdout = ""
    for stage in ("headline", "headline+cpu+traffic", "final"):
        stdout += compact_line(full, stage) + "\n"
    last = stdout[-8000:].splitlines()[-1]
    assert len(last) < LINE_BUDGET <= 6000
    rec = json.loads(last)
    for k in CONTRACT:
        assert k in rec, k
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]
    assert "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms_avg",
              "algorithmic_bytes_per_launch"):
        assert k in roof, k
    assert roof["achieved"] == full["roofline"]["achieved"] and roof["frac"] == full["roofline"]["frac"]
    assert abs(roof["traffic"] / full["roofline"]["traffic"] - 1) < 1e-6
    cpu = rec["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample", "value_1core"} <= set(cpu) and cpu["kind"] == "port"
    assert rec["kernel_only_value"] == full["kernel_only_value"]
    # every secondary leg has its short record, with the consumer-side calls
    ss = rec["secondary_summary"]
    for leg in full["secondary"]:
        if isinstance(full["secondary"][leg], dict):
            assert leg in ss, leg
    assert ss["device_loop_q2hex"]["calls"]["internal_force"][0] > 0
    assert ss["mohr_coulomb_cfg4"]["bound"] == "fp64_valu" and 0 < ss["mohr_coulomb_cfg4"]["frac"] < 1
    # every earlier line of the same run is a valid contract line too
    for ln in stdout.splitlines():
        r = json.loads(ln)
        assert all(k in r for k in CONTRACT) and len(ln) < LINE_BUDGET


def test_headline_stage_without_side_legs():
    full = canned()
    for k in ("cpu_baseline", "end_to_end", "secondary"):
        full.pop(k)
    full["roofline"].update(traffic=None, traffic_detail=None, traffic_over_algorithmic=None)
    rec = json.loads(compact_line(full, "headline"))
    assert rec["line"] == "headline" and rec["roofline"]["traffic"] is None and "cpu_baseline" not in rec
    assert rec["roofline"]["frac"] == full["roofline"]["frac"]


def test_inflated_record_is_cut_down_never_the_contract_keys():
    full = canned()
    for i in range(40):      # forty more device-loop legs: far beyond the budget
        full["secondary"][f"extra_leg_{i}"] = copy.deepcopy(full["secondary"]["device_loop_q2hex"])
    line = compact_line(full)
    assert len(line) <= LINE_BUDGET
    rec = json.loads(line)
    assert rec.get("truncated") is True
    for k in CONTRACT:
        assert k in rec, k


def test_n_gt_1_record_with_gather_modes_and_check():
    full = canned()
    for k in ("end_to_end", "secondary"):
        full.pop(k)
    full.update(n_gpus=8)
    full["config"].update(gather="rccl_all_gather_compact", sharding="cell-block", rccl_ranks=8, collective_backend="nccl",
                          gather_modes={m: {"value": 1e10, "ms_per_step": 9.5, "link_bytes_per_qp": 56} for m in
                                        ("compact", "compact_pipelined", "compact_direct", "full")},
                          gather_modes_meaning={m: "x" * 400 for m in ("compact", "full")},
                          mode_status={m: "timed" for m in ("compact", "compact_pipelined", "compact_direct", "full")})
    full["gather_check"] = {"status": "ok", "rccl_ranks_in_libdxo": 8, "full_ms_per_step": 1.0, "compact_ms_per_step": 0.5,
                            "compact_replicas_bit_identical": True, "why": "y" * 1000}
    line = compact_line(full, "final+gather_check")
    rec = json.loads(line)
    assert len(line) < LINE_BUDGET
    assert set(rec["config"]["gather_modes"]) == {"compact", "compact_pipelined", "compact_direct", "full"}
    assert "gather_modes_meaning" not in rec["config"]
    assert rec["gather_check"]["status"] == "ok" and len(rec["gather_check"]["why"]) <= 120


def test_batch_timing_keys_reach_the_line():
    """SURVEY 8d's protocol (median of 5 batches): the per-batch step and kernel times and the largest step gap are on the line."""
    full = canned()
    full.update(batches=5, ms_per_step_batches=[0.7251234567, 0.7249, 0.8181, 0.7253, 0.7250], kernel_ms_batches=[0.7181234567, 0.718, 0.7179, 0.7182, 0.718],
                step_gap_us_max=1934.56789,
                timing={"protocol": "p" * 500, "median_batch": 0, "step_gap_us_max_at": {"batch": 2, "after_step": 7}, "step_gap_us_median": 5.9,
                        "step_gap_meaning": "m" * 300})
    for stage in ("headline", "headline+cpu", "headline+cpu+traffic", "final"):
        line = compact_line(full, stage)
        assert len(line) < LINE_BUDGET
        rec = json.loads(line)
        assert rec["batches"] == 5 and len(rec["ms_per_step_batches"]) == 5 and len(rec["kernel_ms_batches"]) == 5
        assert abs(rec["ms_per_step_batches"][2] - 0.8181) < 1e-9 and abs(rec["step_gap_us_max"] - 1934.6) < 0.1
        assert rec["timing"]["median_batch"] == 0 and rec["timing"]["step_gap_us_max_at"] == {"batch": 2, "after_step": 7}
        assert len(rec["timing"]["protocol"]) < 200 and "step_gap_meaning" not in rec["timing"]
    # the inflated record keeps them too: they are never among the parts dropped to fit the budget
    for i in range(40):
        full["secondary"][f"extra_leg_{i}"] = copy.deepcopy(full["secondary"]["device_loop_q2hex"])
    rec = json.loads(compact_line(full))
    assert rec.get("truncated") is True and len(rec["ms_per_step_batches"]) == 5 and "step_gap_us_max" in rec


def test_compact_record_is_plain_json_types():
    rec = compact_record(canned())
    json.loads(json.dumps(rec, allow_nan=False))      # no NaN / Infinity tokens a strict parser would refuse
