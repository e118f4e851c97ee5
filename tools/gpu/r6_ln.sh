#!/bin/bash
TAG=${1:-r09e}
O=gpurun_out/$TAG
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round6.py -x -q > $O/round6_tests.log 2>&1; tail -15 $O/round6_tests.log
for r in 0 RULE 2 3 4 6 8 0 RULE; do
  if [ "$r" = RULE ]; then timeout 100 python tools/probe_layer_norm.py >> $O/layer_norm_rows.txt 2>&1; else RTEN_LN_ROWS=$r timeout 100 python tools/probe_layer_norm.py >> $O/layer_norm_rows.txt 2>&1; fi
done
cat $O/layer_norm_rows.txt
