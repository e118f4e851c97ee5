#!/bin/bash
# Round-2 kernel iteration: selected GPU tests, then the f32 bench with the per-layer autotune table.
# Usage: gpurun --timeout 1800 -- 'bash tools/gpu/r2b.sh <tag> "<pytest args>" [bench extra args]'
TAG=${1:-r2b}
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest $2 -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 900 python bench.py --layer-table --save-plan gpurun_out/${TAG}_plan.json --no-secondary $3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?" >> gpurun_out/${TAG}_bench.err
tail -25 gpurun_out/${TAG}_pytest.log; head -c 2500 gpurun_out/${TAG}_bench.json; echo; grep "^\[layer\]" gpurun_out/${TAG}_bench.err | sed -e 's/us:.*| split/| split/' | head -60; tail -3 gpurun_out/${TAG}_bench.err
