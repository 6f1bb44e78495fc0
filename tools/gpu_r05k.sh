#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -2;  timeout 300 python tools/train_slice_bench.py --full --height 320 --width 736 --batch 1 2>&1 | tail -1; timeout 300 python tools/train_slice_bench.py 2>&1 | tail -1 ) > gpurun_out/train_full.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r05z_train -- python "$REPO/tools/train_slice_bench.py" --full --steps 4 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_train_full.log"
cd "$REPO"
db=$(find /tmp/prof_train -name "r05z_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r05z_train_kernel_stats.txt > /dev/null
cat gpurun_out/train_full.log; head -40 gpurun_out/r05z_train_kernel_stats.txt | cut -c1-200
