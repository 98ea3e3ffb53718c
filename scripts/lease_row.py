import json, sys
d = json.load(open(sys.argv[1]))
s = d["secondary"]
hx, tr = s["device_loop_q2hex"], s["device_loop_p2tri"]
r = d["roofline"]
cb = d.get("cpu_baseline", {})
print(" | ".join(str(x) for x in [sys.argv[2], round(d["value"] / 1e9, 4), round(r["frac"], 4), round(r["kernel_ms_avg"], 4), round(r.get("traffic_over_algorithmic") or 0, 4),
      round(r.get("achieved_plain_hipMalloc") or 0, 1), round(cb.get("value", 0), 1), cb.get("cores"), round(s["mohr_coulomb_cfg4"]["ms_per_launch"], 4), round(s["icnn_cfg5"]["ms_per_launch"], 4),
      round(s["vm_field_q2"]["ms_per_launch"], 4), round(hx["iteration_ms"], 4), round(hx["without_tangent_array"]["iteration_ms"], 4), round(tr["iteration_ms"], 4),
      round(tr["without_tangent_array"]["iteration_ms"], 4), d.get("wall_s")]))
