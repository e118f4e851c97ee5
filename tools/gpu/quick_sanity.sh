#!/bin/bash
# one-minute sanity of the final tree: smoke, the chained-runner / shared-context / split-K fold tests, a short default bench line
mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -k "chained or two_threads or last_arrival or split_k" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 10 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['batch_chains']['placement'], d['roofline']['step']['frac'])"
