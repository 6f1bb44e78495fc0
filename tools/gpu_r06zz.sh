#!/bin/bash
# round 6, last call: the whole GPU suite + smoke + the driver's bench command + the training step with its trace, on the final tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/e2e_stats.jsonl gpurun_out/chain_stats.jsonl
( timeout 2400 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -200 ) > gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -20 ) > gpurun_out/smoke.log
( timeout 900 python bench.py 2>&1 | tail -5 ) > gpurun_out/bench.log
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -1; timeout 300 python tools/train_slice_bench.py 2>&1 | tail -1 ) > gpurun_out/train_step.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o r06zz_train -- python "$REPO/tools/train_slice_bench.py" --full --steps 4 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_train.log"
cd "$REPO"
db=$(find /tmp/prof_train -name "r06zz_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/r06zz_train_kernel_stats.txt > /dev/null
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-300 gpurun_out/bench.log | tail -1; cat gpurun_out/train_step.log | cut -c1-200; head -8 gpurun_out/r06zz_train_kernel_stats.txt | cut -c1-140
