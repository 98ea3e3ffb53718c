"""The result line bench.py writes to stdout, kept SHORT.

Round 4's line was the whole record (38 KB: every leg's workload text, methods, thread scans, per-candidate placement
rates) and the driver could not read it back. The contract line is now a bounded extract of the full record:

    compact_line(full)  ->  one JSON line, <= LINE_BUDGET bytes (asserted), carrying the contract keys (metric, value, unit,
                            n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config with
                            `workload`, roofline, cpu_baseline) the `timing` block (per-batch step and kernel times, largest step gap) plus one short record per secondary leg.

The full record goes to `bench_full.json` beside bench.py and to stderr (bench.py does that); nothing here touches the GPU
or imports torch, so tests/test_bench_line.py builds the line from canned records on the CPU.
"""
from __future__ import annotations

import json

LINE_BUDGET = 6000     # bytes; the driver keeps an 8 KB tail of stdout — the whole last line must fit inside it


def _r(x, sig=6):
    """Round a float to `sig` significant digits (summaries only; `value` and `ms_per_step` keep every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float(f"{x:.{sig}g}")


def _pick(src, keys, sig=6):
    return {k: _r(src[k], sig) for k in keys if isinstance(src, dict) and k in src and src[k] is not None}


def _roof_short(roof):
    """[bound, frac, traffic / algorithmic] of one roofline object (nested `hbm` companion appended when there is one)."""
    if not isinstance(roof, dict):
        return None
    out = {"bound": roof.get("bound"), "frac": _r(roof.get("frac"), 4)}
    toa = roof.get("traffic_over_algorithmic")
    if toa is not None:
        out["toa"] = _r(toa, 4)
    hbm = roof.get("hbm")
    if isinstance(hbm, dict):
        out["hbm_frac"] = _r(hbm.get("frac"), 4)
        if hbm.get("traffic_over_algorithmic") is not None:
            out["toa"] = _r(hbm["traffic_over_algorithmic"], 4)
    useful = roof.get("useful")
    if isinstance(useful, dict) and useful.get("frac") is not None:
        out["useful_frac"] = _r(useful["frac"], 4)
    return out


def _leg_ms(leg):
    for k in ("ms_per_launch", "ms_per_call", "iteration_ms", "ms_per_step", "ms"):
        if isinstance(leg.get(k), (int, float)):
            return k, leg[k]
    return None, None


def _calls_short(calls):
    """{call: [ms, frac, traffic/algorithmic]} — the per-call figures of a device-loop leg."""
    out = {}
    for name, c in (calls or {}).items():
        if not isinstance(c, dict):
            continue
        roof = c.get("roofline") or {}
        out[name] = [_r(c.get("ms_per_call"), 4), _r(roof.get("frac"), 3), _r(roof.get("traffic_over_algorithmic"), 3)]
    return out


def leg_summary(leg):
    """One short record per secondary leg: time, the roof that bounds it, fraction, wasted-traffic ratio, CPU port's rate."""
    if not isinstance(leg, dict):
        return None
    if "error" in leg and len(leg) <= 3:
        return {"error": str(leg["error"])[:80]}
    out = {}
    k, ms = _leg_ms(leg)
    if k:
        out["ms"] = _r(float(ms), 5)
    if isinstance(leg.get("value"), (int, float)):
        out["value"] = _r(float(leg["value"]), 5)
    for k in ("us_per_call", "us_per_step", "us_per_step_bind", "us_per_step_three_calls"):      # the latency-bound legs (demo sizes)
        if isinstance(leg.get(k), (int, float)):
            out[k] = _r(float(leg[k]), 4)
    rs = _roof_short(leg.get("roofline"))
    if rs:
        out.update(rs)
    cpu = leg.get("cpu_baseline")
    if isinstance(cpu, dict) and cpu.get("value") is not None:
        out["cpu"] = _r(float(cpu["value"]), 4)
        if cpu.get("cores") is not None:
            out["cpu_cores"] = cpu["cores"]
        for k in ("us_per_call", "us_per_step"):
            if isinstance(cpu.get(k), (int, float)):
                out["cpu_" + k] = _r(float(cpu[k]), 4)
    if isinstance(leg.get("calls"), dict):
        out["calls"] = _calls_short(leg["calls"])
    wo = leg.get("without_tangent_array")
    if isinstance(wo, dict):
        out["no_tangent_ms"] = _r(wo.get("iteration_ms"), 5)
        out["no_tangent_calls"] = _calls_short(wo.get("calls"))
    plan = leg.get("plan")
    if isinstance(plan, dict) and plan.get("ms_per_call") is not None:
        out["plan"] = [_r(plan["ms_per_call"], 4), _r((plan.get("roofline") or {}).get("frac"), 3),
                       _r((plan.get("roofline") or {}).get("traffic_over_algorithmic"), 3)]
    return out


def secondary_summary(secondary):
    if not isinstance(secondary, dict):
        return None
    out = {}
    for name, leg in secondary.items():
        if isinstance(leg, dict):
            s = leg_summary(leg)
            if s:
                out[name] = s
    out["_keys"] = "ms per launch/call/iteration; frac of the named roof (8 TB/s HBM, 78.6 TF fp64 VALU, 2.5 PF bf16 MFMA); " \
                   "toa = counter HBM bytes / algorithmic; cpu = C port qp/s; calls = [ms, frac, toa]"
    return out


def end_to_end_summary(e2e):
    """The host-array (PCIe-inclusive) rates, never `value`: qp/s per form at each size, dispatcher ms per call."""
    if not isinstance(e2e, dict):
        return None
    out = {}
    for entry in e2e.get("sizes", []):
        rec = {name: _r(float(entry[name]["qp_per_s"]), 4) for name in ("copy", "rebuild", "resident") if name in entry}
        td = entry.get("through_dispatcher")
        if isinstance(td, dict):
            rec["dispatcher_ms"] = {k: td[k] for k in ("default", "bind", "bind_resident_state") if k in td}
        out[str(entry.get("points"))] = rec
    out["unit"] = "qp/s, NumPy in / NumPy out through the C ABI (H2D + kernel + D2H)"
    return out


_CONFIG_KEYS = ("workload", "points_per_gpu", "cells_per_gpu", "nq", "d", "sharding", "gather", "rccl_ranks", "collective_backend",
                "gather_in_place", "placement_candidates_probed", "kernel", "arch", "compute_units")
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms_avg",
              "algorithmic_bytes_per_launch", "bytes_per_qp", "achieved_plain_hipMalloc", "achieved_factory_device_call_arena_outputs",
              "achieved_factory_default_device_call", "stream_probe_GBps")
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "dry_collective", "degraded", "note")


def compact_record(full, stage="final"):
    """The bounded extract of a full bench record (a dict; see compact_line for the string)."""
    out = {k: full[k] for k in _TOP_KEYS if k in full}
    cfg = full.get("config") or {}
    c = {k: cfg[k] for k in _CONFIG_KEYS if cfg.get(k) is not None}
    modes = cfg.get("gather_modes")
    if isinstance(modes, dict):
        c["gather_modes"] = {m: _pick(v, ("value", "ms_per_step", "link_bytes_per_qp")) for m, v in modes.items()}
        if cfg.get("mode_status"):
            c["mode_status"] = {m: ("timed" if s == "timed" else "failed") for m, s in cfg["mode_status"].items()}
    out["config"] = c
    # the timing protocol's evidence: every batch's step time and kernel-event time, the largest gap between two launches
    if full.get("batches") is not None:
        out["batches"] = full["batches"]
    for k in ("ms_per_step_batches", "kernel_ms_batches"):
        if isinstance(full.get(k), list):
            out[k] = [_r(float(x), 6) for x in full[k]]
    if full.get("step_gap_us_max") is not None:
        out["step_gap_us_max"] = _r(float(full["step_gap_us_max"]), 5)
    tm = full.get("timing")
    if isinstance(tm, dict):
        t = _pick(tm, ("median_batch", "step_gap_us_median"), 5)
        if isinstance(tm.get("step_gap_us_max_at"), dict):
            t["step_gap_us_max_at"] = tm["step_gap_us_max_at"]
        t["protocol"] = "warm-up, then `batches` batches of `steps` steps, each between fences; value and ms_per_step = the median batch"
        out["timing"] = t
    roof = full.get("roofline") or {}
    r = {k: (roof[k] if k in ("bound", "unit", "kernel", "achieved", "frac", "kernel_ms_avg") else _r(roof[k], 7))
         for k in _ROOF_KEYS if k in roof}
    r.setdefault("traffic", None)
    td = roof.get("traffic_detail")
    if isinstance(td, dict):
        r["traffic_detail"] = _pick(td, ("fetch_bytes", "write_bytes", "launches_averaged"), 7)
        r["traffic_method"] = "rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate child passes of this run); x1024, FETCH x2 (gfx950)"
    elif isinstance(td, str):
        r["traffic_detail"] = td[:160]
    pl = roof.get("placement")
    if isinstance(pl, dict):
        r["placement"] = _pick(pl, ("mode", "candidates", "chosen_kind", "chosen_GBps", "calibration_ms", "rounds"))
        gb = pl.get("probe_GBps")
        if isinstance(gb, list) and gb:
            r["placement"]["probe_GBps_min_max"] = [min(gb), max(gb)]
    fp = roof.get("factory_placement")
    if isinstance(fp, dict):
        r["factory_placement"] = _pick(fp, ("candidates", "chosen_kind", "chosen_GBps", "rounds"))
    out["roofline"] = r
    cpu = full.get("cpu_baseline")
    if isinstance(cpu, dict):
        cb = _pick(cpu, ("value", "unit", "cores", "kind", "value_1core"), 7)
        cb["sample"] = str(cpu.get("sample", ""))[:220]
        out["cpu_baseline"] = cb
    if "kernel_only_value" in full:
        out["kernel_only_value"] = full["kernel_only_value"]
    if full.get("end_to_end") is not None:
        out["end_to_end_summary"] = end_to_end_summary(full["end_to_end"])
    if full.get("secondary") is not None:
        out["secondary_summary"] = secondary_summary(full["secondary"])
    gc = full.get("gather_check")
    if isinstance(gc, dict):
        out["gather_check"] = {k: (v if not isinstance(v, str) else v[:120]) for k, v in gc.items()
                               if k in ("status", "why", "rccl_ranks_in_libdxo", "full_ms_per_step", "compact_ms_per_step",
                                        "direct_ms_per_step", "overlap_ms_per_step", "compact_replicas_bit_identical",
                                        "library_full_size_ms_per_step")}
    out["line"] = stage
    out["full_record"] = "bench_full.json (beside bench.py) and stderr"
    return _finite(out)


def _finite(o):
    """NaN / Infinity are not JSON: a strict parser refuses the whole line. They become null."""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    return o


def compact_line(full, stage="final", budget=LINE_BUDGET):
    """One JSON line <= budget bytes. If a record ever outgrows the budget, the least important parts are dropped in a
    fixed order (never a contract key) until it fits; the final assert is the guarantee the driver relies on."""
    rec = compact_record(full, stage)
    drops = [("secondary_summary", "_keys"), ("roofline", "placement"), ("roofline", "factory_placement"), ("roofline", "traffic_method"), ("end_to_end_summary", None),
             ("config", "gather_modes"), ("secondary_summary", None), ("gather_check", None)]
    line = json.dumps(rec, separators=(",", ":"), allow_nan=False)
    # per-call arrays of the device-loop legs go first, one leg at a time
    if len(line) > budget and isinstance(rec.get("secondary_summary"), dict):
        for leg in rec["secondary_summary"].values():
            if isinstance(leg, dict):
                for k in ("no_tangent_calls", "calls"):
                    if k in leg and len(line) > budget:
                        del leg[k]
                        line = json.dumps(rec, separators=(",", ":"))
    for top, sub in drops:
        if len(line) <= budget:
            break
        if sub is None:
            rec.pop(top, None)
        elif isinstance(rec.get(top), dict):
            rec[top].pop(sub, None)
        rec["truncated"] = True
        line = json.dumps(rec, separators=(",", ":"))
    if len(line) > budget:   # cannot happen with the contract keys alone (< 2 KB); fail loudly rather than print an unreadable line
        raise AssertionError(f"bench result line is {len(line)} bytes (> {budget})")
    return line
