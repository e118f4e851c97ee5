#!/bin/bash
# the batch of 32 as N independent sub-batch chains (tools/probe_two_chains.py): gpurun --timeout 900 -- 'bash tools/gpu/r2_chains.sh <tag> [chains] [sizes]'
tag=${1:-r2t}
mkdir -p gpurun_out
if [ -n "$3" ]; then extra="--sizes $3"; fi
timeout 600 python tools/probe_two_chains.py --steps 50 --chains ${2:-1,2,4} $extra > gpurun_out/${tag}_chains.txt 2> gpurun_out/${tag}_chains.err
cat gpurun_out/${tag}_chains.txt | cut -c1-220; tail -3 gpurun_out/${tag}_chains.err
