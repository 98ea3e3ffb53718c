#!/bin/bash
# quick check of the operand kernels on the GPU box: tests, then timings
python3 -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 900 python3 -m pytest tests/test_operand_eval.py -x -q -m gpu 2>&1 | tail -4
python3 scripts/bench_operand.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    print(j['case'][:70], {k: round(v, 4) for k, v in j.items() if k.endswith('_ms')})
"
