#!/bin/bash
# round 6, call P: the training step as it stands (whole model, 512x256 batch 2) + its kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r06p}
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -1 ) > gpurun_out/${TAG}_train_step.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o ${TAG}_train -- python "$REPO/tools/train_slice_bench.py" --full --steps 4 2>&1 | tail -3 ) > "$REPO/gpurun_out/${TAG}_rocprof_train.log"
cd "$REPO"
db=$(find /tmp/prof_train -name "${TAG}_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_train_kernel_stats.txt > /dev/null
cat gpurun_out/${TAG}_train_step.log; head -45 gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-150
