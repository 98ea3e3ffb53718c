"""One row of profiles/r0N_bench_by_lease.txt from a bench_full.json: python scripts/lease_row.py bench_full.json <label>"""
import json, sys
d = json.load(open(sys.argv[1]))
s = d.get("secondary", {})
hx, tr = s.get("device_loop_q2hex", {}), s.get("device_loop_p2tri", {})
r = d["roofline"]
cb = d.get("cpu_baseline", {})
fp = r.get("factory_placement") or {}
fac = r.get("achieved_factory_device_call_arena_outputs") or 0.0


def g(x, *ks, nd=4):
    for k in ks:
        x = (x or {}).get(k)
    return round(x, nd) if isinstance(x, (int, float)) else None


print(" | ".join(str(x) for x in [
    sys.argv[2], round(d["value"] / 1e9, 4), round(d["ms_per_step"], 4), round(r["kernel_ms_avg"], 4), round(r["frac"], 4),
    "[" + " ".join(f"{x:.4f}" for x in d.get("ms_per_step_batches", [])) + "]", round(d.get("step_gap_us_max") or 0, 1),
    round(r.get("traffic_over_algorithmic") or 0, 4), round(r["achieved"], 0), round(fac, 0), round(fac / r["achieved"], 3) if fac else None,
    f"rounds {r['placement'].get('rounds')}/{fp.get('rounds')}", round(r.get("achieved_plain_hipMalloc") or 0, 0),
    round(cb.get("value", 0) / 1e8, 2), cb.get("cores"), g(s, "mohr_coulomb_cfg4", "ms_per_launch"), g(s, "icnn_cfg5", "ms_per_launch"), g(s, "vm_field_q2", "ms_per_launch"),
    g(hx, "iteration_ms"), g(hx, "without_tangent_array", "iteration_ms"), g(tr, "iteration_ms"), g(tr, "without_tangent_array", "iteration_ms"),
    g(s, "heat_cfg1", "us_per_step", nd=1), g(s, "von_mises_demo_host", "us_per_call", nd=1), d.get("wall_s")]))
