#!/bin/bash
# HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes) for the secondary kernels of scripts/bench_extra.py.
set -u
OUT=gpurun_out/${1:-extra_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -o ex -- python3 scripts/bench_extra.py > "$OUT/fetch.log" 2>&1; echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -o ex -- python3 scripts/bench_extra.py > "$OUT/write.log" 2>&1; echo "write rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for name, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    for f in glob.glob(out + "/prof_*/**/ex_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "anonymous namespace" in r["Kernel_Name"]:
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                acc[(k, int(r["Grid_Size"]))][name].append(float(r["Counter_Value"]) * 1024 * corr)
rows = []
for (k, grid), cs in sorted(acc.items()):
    rd = sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1)
    wr = sum(cs["WRITE_SIZE"]) / max(len(cs["WRITE_SIZE"]), 1)
    rows.append({"kernel": k, "grid_threads": grid, "read_MB": round(rd / 1e6, 1), "write_MB": round(wr / 1e6, 1), "total_MB": round((rd + wr) / 1e6, 1)})
    print(rows[-1])
json.dump(rows, open(out + "/extra_pmc_summary.json", "w"), indent=1)
PY
find "$OUT" -name "*.db" -delete
