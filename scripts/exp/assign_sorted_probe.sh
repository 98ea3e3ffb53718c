#!/bin/bash
# scripts/exp/assign_sorted_probe.sh — run the probe and its two counter passes on the GPU box (repo root).
mkdir -p gpurun_out/assign6
export TMPDIR=/tmp
Z=${1:-0}      # 1: the bench leg's numbering (z fastest)
scripts/exp/assign_sorted_probe 108 $Z | tee gpurun_out/assign6/probe_$Z.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/assign6/$c -o p -- scripts/exp/assign_sorted_probe 108 $Z > /dev/null 2>&1
  python3 - "$c" <<'PY' | tee -a gpurun_out/assign6/probe_$Z.txt
import csv, glob, collections, sys
c = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/assign6/{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "apply_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(c, k, round(sum(v) / len(v) * 1024 / 1e6, 1), "MB per launch (raw counter x 1024)")
PY
  rm -rf gpurun_out/assign6/$c
done
