#!/bin/bash
tag=${1:-r2t}
mkdir -p gpurun_out
{
for i in 1 2; do timeout 300 python tools/probe_two_chains.py --steps 50 --chains 4; timeout 300 python tools/probe_two_chains.py --steps 50 --chains 4 --stagger; done
timeout 300 python tools/probe_two_chains.py --steps 50 --chains 2 --stagger
} > gpurun_out/${tag}_chains.txt 2> gpurun_out/${tag}_chains.err
cat gpurun_out/${tag}_chains.txt | cut -c1-220; tail -3 gpurun_out/${tag}_chains.err
