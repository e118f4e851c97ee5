R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bert -o bert -- python $R/tools/bench_bert.py > $R/gpurun_out/prof_bert.json 2> $R/gpurun_out/prof_bert.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_i8 -o i8 -- python $R/tools/bench_resnet50_int8.py > $R/gpurun_out/prof_i8.json 2> $R/gpurun_out/prof_i8.err
ls $R/gpurun_out/prof_bert $R/gpurun_out/prof_i8
