#!/bin/bash
# Round 4: row-wise / pooling / depthwise kernel pass: parity, then the microbenchmarks of the touched operators.
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_einsum.py tests/test_gpu_round3.py -m gpu -x -q -k "pool or depthwise or reduce_sum or einsum" 2>&1 | tail -4
for op in ReduceSum MaxPool depthwise GlobalAveragePool Softmax; do timeout 120 python tools/bench_ops.py --only "$op" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        try:
            d = json.loads(l)
            for r in d.get('rows', [d]): print(r.get('op'), r.get('shape'), r.get('us'), r.get('frac'))
        except Exception as e: print('?', l[:200])
"; done
