#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 600 python tools/train_slice_bench.py 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r05f_train_step.txt
( timeout 600 python tools/train_slice_bench.py --height 320 --width 736 --batch 1 2>&1 | grep -v amdgpu.ids | tail -1 ) >> gpurun_out/r05f_train_step.txt
cat gpurun_out/r05f_train_step.txt
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r05f_train -- python "$GRAFT_REPO_ROOT/tools/train_slice_bench.py" --steps 3 2>&1 | tail -2 ) > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_train.log"
cd "$GRAFT_REPO_ROOT"
db=$(find /tmp/prof_train -name "r05f_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r05f_train_kernel_stats.txt > /dev/null
head -16 gpurun_out/r05f_train_kernel_stats.txt | cut -c1-150
