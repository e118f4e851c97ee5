#!/bin/bash
run() { timeout 120 python tools/bench_ops.py --only "$1" 2>/dev/null | python -c "
import sys, json
t = sys.stdin.read()
d = json.loads(t[t.index('{'):])
for r in d['rows']: print(r.get('op'), r.get('shape'), r.get('us'), r.get('frac'))
"; }
run ReduceSum
timeout 200 python -m pytest tests/test_einsum.py -m gpu -x -q -k "reduce_sum" 2>&1 | tail -2
