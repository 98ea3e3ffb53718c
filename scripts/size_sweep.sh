#!/bin/bash
# bench.py over batch sizes (GPU box): 1e5 ... 1e8 points per GPU, d = 6 and the d = 4 reference layout.
OUT=gpurun_out/${1:-sweep}
mkdir -p "$OUT"
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
: > "$OUT/size_sweep.jsonl"
for d in 6 4; do
  for n in 100000 1000000 10000000 100000000; do
    steps=20; [ "$n" -ge 100000000 ] && steps=10
    timeout 900 python3 bench.py --nqp $n --d $d --steps $steps --no-cpu --placement 4 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.readline())
r = j['roofline']
print(json.dumps({'d': $d, 'points': j['config']['points_per_gpu'], 'value_qp_per_s': j['value'], 'ms_per_step': j['ms_per_step'],
                  'kernel_GBps': r['achieved'], 'frac_of_8TBps': r['frac'], 'first_allocation_GBps': r['achieved_first_allocation'],
                  'stream_ceiling_GBps': r['stream_ceiling_GBps']}))
" | tee -a "$OUT/size_sweep.jsonl"
  done
done
