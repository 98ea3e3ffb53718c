#!/bin/bash
# Counter passes for the ICNN MFMA kernel (GPU box, repo root). Separate --pmc passes, no tracing flags.
set -u
OUT=gpurun_out/${1:-icnn_prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o ic -- python3 scripts/bench_icnn.py --launches 2 --variant ${ICNN_VARIANT:-2} > "$OUT/$name.log" 2>&1; echo "$name rc=$?"
}
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
pass p3 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM
pass p4 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/ic_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "icnn_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: round(sum(v) / len(v), 1) for c, v in sorted(acc.items())})
PY
find "$OUT" -name "*.db" -delete
