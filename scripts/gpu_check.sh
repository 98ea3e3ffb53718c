#!/bin/bash
# One gpurun call: build, GPU tests, smoke, bench, rocprofv3 kernel stats + HBM counter passes.
# usage (from the repo root, on the GPU box):  bash scripts/gpu_check.sh [tag]
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import __graft_entry__ as g; g.build()" > "$OUT/build.log" 2>&1 || { echo BUILD FAILED; tail -30 "$OUT/build.log"; exit 1; }
echo "== pytest -m gpu"
timeout 1500 python3 -m pytest tests -x -q -m gpu --durations=12 > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -30 "$OUT/pytest_gpu.log"
echo "== smoke"
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== bench"
timeout 900 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cat "$OUT/bench.json"; tail -5 "$OUT/bench.err"
echo "== rocprofv3 kernel-trace --stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o vm -- python3 bench.py --no-cpu --no-probe --no-e2e --no-traffic > "$OUT/prof_stats.log" 2>&1; echo "rocprof rc=$?"
find "$OUT/prof_stats" -name "*kernel_stats.csv" | head -1 | xargs -r head -12
echo "== rocprofv3 --pmc FETCH_SIZE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch" -o vm -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-probe --no-e2e --no-traffic > "$OUT/prof_fetch.log" 2>&1; echo "rc=$?"
echo "== rocprofv3 --pmc WRITE_SIZE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/prof_write" -o vm -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-probe --no-e2e --no-traffic > "$OUT/prof_write.log" 2>&1; echo "rc=$?"
python3 scripts/bench_extra.py > "$OUT/bench_extra.jsonl" 2>&1; cat "$OUT/bench_extra.jsonl"
python3 scripts/summarize_pmc.py "$OUT" --json "$OUT/traffic.json" > "$OUT/pmc_summary.txt" 2>&1; cat "$OUT/pmc_summary.txt"
# keep the merged directory small
find "$OUT" -name "*.db" -size +20M -delete
du -sh "$OUT"
