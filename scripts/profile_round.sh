#!/bin/bash
# One gpurun call that produces every file profiles/ needs for a round (GPU box, repo root):
#   bash scripts/profile_round.sh <tag> [quick]
#     bench.json                 the full record of the default `python3 bench.py` run (bench_full.json); bench_stdout.txt = the bounded lines the
#                                driver sees, <tag>_bench_line.json = the last of them
#     stats/                     rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu --no-probe --no-e2e` (headline AND
#                                secondary kernels: vm_tile, mc_fused, icnn_mfma_bf16x3 / icnn_mfma_f32, vm_field, ...)
#     prof_fetch/, prof_write/   HBM counters of the headline kernel, separate --pmc passes (--no-secondary)
#     mc_*/ icnn_*/ field_*/     counter passes of the secondary kernels (skipped with `quick`)
# then scripts/summarize_round.py writes the text / JSON summaries next to them; copy those into profiles/.
set -u
TAG=${1:-r04}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { echo BUILD FAILED; tail -30 "$OUT/build.log"; exit 1; }
S0=$(date +%s)
timeout 900 python3 bench.py > "$OUT/bench_stdout.txt" 2> "$OUT/bench.err"; echo "bench rc=$? wall=$(( $(date +%s) - S0 )) s, stdout $(wc -c < "$OUT/bench_stdout.txt") bytes in $(wc -l < "$OUT/bench_stdout.txt") lines"
cp bench_full.json "$OUT/bench.json" 2>/dev/null        # the full record (the stdout lines are bounded extracts of it)
tail -1 "$OUT/bench_stdout.txt" > "$OUT/${TAG}_bench_line.json"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- python3 bench.py --no-cpu --no-probe --no-e2e --no-traffic --no-side > "$OUT/bench_profiled.json" 2> "$OUT/stats.log"; echo "kernel-trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -o vm -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-probe --no-e2e --no-secondary --no-traffic --no-side > "$OUT/prof_fetch.log" 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -o vm -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-probe --no-e2e --no-secondary --no-traffic --no-side > "$OUT/prof_write.log" 2>&1; echo "write rc=$?"
if [ -z "$QUICK" ]; then
  pass() {  # dir, script + args..., -- counters...
    local name=$1; shift
    local cmd=(); while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
    timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o k -- python3 "${cmd[@]}" > "$OUT/$name.log" 2>&1; echo "$name rc=$?"
  }
  MC="scripts/bench_mc.py --launches 2 --variant 1"     # two-kernel form: the flop count per plastic point / per classified point
  pass mc_valu $MC -- SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES
  pass mc_wave $MC -- SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pass mc_fetch $MC -- FETCH_SIZE
  pass mc_write $MC -- WRITE_SIZE
  MCF="scripts/bench_mc.py --launches 2 --variant 2"    # the default single persistent kernel
  pass mcf_valu $MCF -- SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES
  pass mcf_wave $MCF -- SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pass mcf_lane $MCF -- SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU     # lane utilisation of the default kernel: thread-cycles / (64 x wave-cycles)
  pass mcf_fetch $MCF -- FETCH_SIZE
  pass mcf_write $MCF -- WRITE_SIZE
  IC="scripts/bench_icnn.py --launches 2"
  pass icnn_p1 $IC -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_INSTS_VALU
  pass icnn_p2 $IC -- SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
  pass icnn_p3 $IC -- SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
  DL="tools/bench_secondary.py --child --legs device_loop_q2hex,device_loop_p2tri,assign_cg"   # the device-resident Newton iteration + assigner
  pass dl_fetch $DL -- FETCH_SIZE
  pass dl_write $DL -- WRITE_SIZE
  pass dl_sq1 $DL -- SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass dl_sq2 $DL -- SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass dl_flop $DL -- SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64     # fp64 work of the consumer-side kernels
  OP="scripts/bench_operand.py --launches 2 --case 0 --operand-cell 0"
  pass field_fetch $OP -- FETCH_SIZE
  pass field_write $OP -- WRITE_SIZE
  pass field_sq1 $OP -- SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES
  pass field_sq2 $OP -- SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE
  pass field_tcc $OP -- TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
fi
python3 scripts/summarize_round.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*_agent_info.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +1M -exec gzip -f {} \;      # gpurun brings back at most 64 MiB; the summaries above are what profiles/ keeps
du -sh "$OUT"
