#!/bin/bash
# Mohr-Coulomb profile: kernel stats + fp64 VALU instruction counters (separate passes). GPU box, repo root.
set -u
OUT=gpurun_out/${1:-mc_r01}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
python3 scripts/bench_mc.py > "$OUT/bench_mc.json" 2>/dev/null; cut -c1-400 "$OUT/bench_mc.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o mc -- python3 scripts/bench_mc.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/pmc_valu" -o mc -- python3 scripts/bench_mc.py --launches 2 > "$OUT/pmc_valu.log" 2>&1; echo "pmc rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_wave" -o mc -- python3 scripts/bench_mc.py --launches 2 > "$OUT/pmc_wave.log" 2>&1; echo "pmc rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/mc_kernel_stats.csv", recursive=True):
    for r in csv.reader(open(f)):
        if "mc_" in r[0]:
            print("stats", r[0].split("(")[1][:40] if "(" in r[0] else r[0][:40], "calls", r[1], "avg_ns", r[3])
for d in ("pmc_valu", "pmc_wave"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + f"/{d}/**/mc_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mc_" in r["Kernel_Name"]:
                k = "newton" if "newton" in r["Kernel_Name"] else ("classify" if "classify" in r["Kernel_Name"] else "point")
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d, k, {c: sum(v) / len(v) for c, v in cs.items()})
PY
find "$OUT" -name "*.db" -delete; du -sh "$OUT"
