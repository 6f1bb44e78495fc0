#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --config4 --no-cpu-baseline --no-stream-figure > gpurun_out/c4_$i.out 2> gpurun_out/c4_$i.err
  echo "run $i rc=$?"; grep -c '^{' gpurun_out/c4_$i.out; tail -5 gpurun_out/c4_$i.err | cut -c1-300
done
