#!/bin/bash
# Counter passes for the operand kernels (GPU box, repo root). Separate --pmc passes, no tracing flags.
set -u
OUT=gpurun_out/${1:-op_prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1
pass() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o op -- python3 scripts/bench_operand.py --launches 2 --case ${CASE:-0} --operand-cell ${OPCELL:-0} > "$OUT/$name.log" 2>&1; echo "$name rc=$?"
}
pass hbm_r FETCH_SIZE
pass hbm_w WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass sq3 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_UTCL1_STALL_MULTI_MISS_sum GRBM_GUI_ACTIVE
pass lds SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/op_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "operand_eval" in k or "vm_field" in k or "vm_tile" in k:
            key = ("operand_eval" if "operand_eval" in k else "vm_field" if "vm_field" in k else "vm_tile") + " grid=" + r.get("Grid_Size", "?")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()})
PY
find "$OUT" -name "*.db" -delete; du -sh "$OUT"
