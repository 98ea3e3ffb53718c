import json,sys
r=json.loads(sys.stdin.read()); pl=r["roofline"]["placement"]
print(round(r["roofline"]["frac"],3), "tuned_bpc", pl.get("tuned_blocks_per_cu"), "plain", round(r["roofline"]["achieved_plain_hipMalloc"] or 0), "chosen", pl["chosen"], pl["chosen_kind"], pl["chosen_GBps"], "cal_ms", round(pl["calibration_ms"]), pl["probe"], [k[0]+str(round(v)) for k,v in zip(pl["kinds"], pl["probe_GBps"])])
