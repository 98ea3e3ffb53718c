#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel.

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B... see MI355X_MICROARCH.md "HBM":
hbm_bytes = counter * 1024, and on gfx950 FETCH_SIZE under-counts a wide coalesced streaming read by 2x
(doubled below, as the guide prescribes). WRITE_SIZE is uncalibrated there; we calibrate it against the
kernel's known algorithmic write bytes in DESIGN.md.
"""
import csv
import pathlib
import sys
from collections import defaultdict

out = pathlib.Path(sys.argv[1])
for pass_name, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    files = list((out / pass_name).rglob("*counter_collection.csv"))
    if not files:
        print(f"{pass_name}: no counter_collection.csv")
        continue
    per = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == counter:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:6]:
        mean = sum(v) / len(v)
        print(f"{counter} {k[:90]:90s} launches={len(v):4d} mean={mean:.1f} -> {mean * 1024 / 1e6:.1f} MB/launch raw")
