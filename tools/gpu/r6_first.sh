#!/bin/bash
# Round 6, first session: the driver's exact command with its raw stdout kept (the line must be <= 4 KB and the last thing printed), then the GPU test suite.
TAG=${1:-r09a}
O=gpurun_out/$TAG
mkdir -p $O
( rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" ) > $O/hardware.txt 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd.stdout 2> $O/driver_cmd.stderr; echo "rc=$?" >> $O/driver_cmd.stderr
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
cp gpurun_out/bench_detail_int8.json $O/bench_detail_int8.json 2>/dev/null
wc -c $O/driver_cmd.stdout $O/driver_cmd.stderr
tail -c 4200 $O/driver_cmd.stdout
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -5 $O/gputests.log
