#!/bin/bash
# compile and run a standalone .hip probe on the GPU box: tools/gpu/r2_hip_probe.sh <tag> <probe.hip>
TAG=$1
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/probe_bin $2 && timeout 300 /tmp/probe_bin > gpurun_out/${TAG}.txt 2>&1
cat gpurun_out/${TAG}.txt
