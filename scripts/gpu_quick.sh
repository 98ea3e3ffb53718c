#!/bin/bash
# Short GPU pass: von Mises tests, single-rank RCCL bench (both gather modes), secondary benches with CPU legs.
set -u
TAG=${1:-quick}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { echo BUILD FAILED; tail -30 "$OUT/build.log"; exit 1; }
timeout 900 python3 -m pytest tests/test_von_mises_gpu.py -x -q > "$OUT/pytest_vm.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest_vm.log"
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --gather 1 --steps 10 --no-cpu > "$OUT/bench_dist1.json" 2> "$OUT/bench_dist1.err"; echo "dist bench rc=$?"; cut -c1-1500 "$OUT/bench_dist1.json"; tail -3 "$OUT/bench_dist1.err"
timeout 600 python3 bench.py --no-cpu > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-300 "$OUT/bench.json"
timeout 900 python3 scripts/bench_extra.py > "$OUT/bench_extra.jsonl" 2>&1; cut -c1-300 "$OUT/bench_extra.jsonl"
timeout 900 python3 scripts/bench_mc.py --cpu 100000 > "$OUT/bench_mc.json" 2> "$OUT/bench_mc.err"; echo "mc rc=$?"; cut -c1-1500 "$OUT/bench_mc.json"
timeout 900 python3 scripts/bench_icnn.py --cpu 20000 > "$OUT/bench_icnn.json" 2> "$OUT/bench_icnn.err"; echo "icnn rc=$?"; cat "$OUT/bench_icnn.json"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_icnn" -o icnn -- python3 scripts/bench_icnn.py > /dev/null 2>&1; echo "rocprof icnn rc=$?"
find "$OUT/prof_icnn" -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-300
find "$OUT" -name "*.db" -size +20M -delete
