#!/bin/bash
# The round's last call: tools/gpu_round.sh (GPU tests, smoke, the driver's bench command, BASELINE configs, kernel traces) + the training step
# (tools/train_slice_bench.py, whole model and slice) with its own kernel trace.   TAG=r05z tools/gpu_final.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd)
TAG=${TAG:-r05z}
LEAN=1 TAG=$TAG tools/gpu_round.sh > gpurun_out/round.log 2>&1
rm -rf gpurun_out/prof                                   # (the kernel-trace databases: summarised above, too large to travel back)
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -1; timeout 300 python tools/train_slice_bench.py --full --height 320 --width 736 --batch 1 2>&1 | tail -1; timeout 300 python tools/train_slice_bench.py 2>&1 | tail -1 ) > gpurun_out/train_step.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o ${TAG}_train -- python "$REPO/tools/train_slice_bench.py" --full --steps 4 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_train.log"
cd "$REPO"
db=$(find /tmp/prof_train -name "${TAG}_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_train_kernel_stats.txt > /dev/null
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-300 gpurun_out/bench.log | tail -2; cat gpurun_out/train_step.log; head -12 gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-140
