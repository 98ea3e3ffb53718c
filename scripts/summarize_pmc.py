#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel and grid size.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": bytes = counter * 1024, and on
gfx950 FETCH_SIZE under-counts a wide coalesced streaming read by exactly 2x (so it is doubled);
WRITE_SIZE is taken as is (it matched the algorithmic write bytes to 4 digits on this kernel).
Writes profiles-ready text to stdout; with --json FILE also writes the traffic record bench.py picks up.
"""
import csv
import json
import pathlib
import sys
from collections import defaultdict

out = pathlib.Path(sys.argv[1])
json_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
rec = {}
for pass_name, counter, corr in (("prof_fetch", "FETCH_SIZE", 2.0), ("prof_write", "WRITE_SIZE", 1.0)):
    files = list((out / pass_name).rglob("*counter_collection.csv"))
    if not files:
        print(f"{pass_name}: no counter_collection.csv")
        continue
    per = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    per[(row["Kernel_Name"], int(row["Grid_Size"]))].append(float(row["Counter_Value"]))
    for (k, grid), v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:8]:
        mean = sum(v) / len(v)
        short = k.replace("(anonymous namespace)::", "")[:70]
        print(f"{counter:10s} grid={grid:10d} launches={len(v):3d} mean={mean:12.1f} KiB -> {mean * 1024 * corr / 1e6:9.1f} MB/launch "
              f"(x{corr:g} gfx950 correction)  {short}")
        if "vm_tile" in k and (counter not in rec or grid > rec[counter][0]):
            rec[counter] = (grid, mean * 1024 * corr)
if json_path and len(rec) == 2 and rec["FETCH_SIZE"][0] == rec["WRITE_SIZE"][0]:
    grid = rec["FETCH_SIZE"][0]
    total = rec["FETCH_SIZE"][1] + rec["WRITE_SIZE"][1]
    pathlib.Path(json_path).write_text(json.dumps({
        "kernel": "vm_tile<6>", "d": 6, "points_per_launch": grid, "grid_threads": grid,
        "fetch_bytes_per_launch": rec["FETCH_SIZE"][1], "write_bytes_per_launch": rec["WRITE_SIZE"][1],
        "hbm_bytes_per_launch": total,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = counter*1024; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 wide-stream under-count)",
    }, indent=1))
    print(f"traffic record: {total / 1e6:.1f} MB per launch at grid {grid}")
