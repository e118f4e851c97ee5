#!/bin/bash
# Round 6, second part: the transformers BERT-base export through the model ABI (arena move + replicas keep the small constants' host mirrors).
TAG=${1:-r10a}
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests/test_shape_arithmetic.py -m gpu -x -q > $O/tests_shape.log 2>&1; tail -3 $O/tests_shape.log
timeout 600 python tools/bench_bert.py --hf --lanes 1 > $O/bench_bert_hf_export_1lane.json 2> $O/bench_bert_hf_export_1lane.err; tail -c 600 $O/bench_bert_hf_export_1lane.err
timeout 600 python tools/bench_bert.py --hf --lanes 4 > $O/bench_bert_hf_export.json 2> $O/bench_bert_hf_export.err; tail -c 600 $O/bench_bert_hf_export.err
tail -c 1500 $O/bench_bert_hf_export_1lane.json; echo; tail -c 1500 $O/bench_bert_hf_export.json
