#!/bin/bash
# Builds librten_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../librten_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result ${RTEN_EXTRA_FLAGS:-}"
mkdir -p ../_build
pids=()
for f in *.hip; do
  o=../_build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ internal.h -nt "$o" ] || [ vecmath.h -nt "$o" ] || [ ../../include/rten_hip.h -nt "$o" ] || \
     { [ "${f#gemm_f32}" != "$f" ] && [ gemm_f32_common.h -nt "$o" ]; }; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -c "$f" -o "$o" &
    pids+=($!)
  fi
done
# host-only translation units (the plan executor of include/rten_hip_graph.hpp behind the C ABI)
for f in *.cpp; do
  o=../_build/${f%.cpp}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ ../../include/rten_hip.h -nt "$o" ] || [ ../../include/rten_hip_graph.hpp -nt "$o" ] || [ ../../include/rten_hip_ops.hpp -nt "$o" ]; then
    g++ -std=c++17 -O2 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../_build/*.o
echo "built $OUT"
