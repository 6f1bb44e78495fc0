#!/bin/bash
# round 6, evidence call: the round script (GPU tests, smoke, the driver's bench command, the other BASELINE configs, kernel traces), the
# training step with its trace, and the --pmc passes (model-level: every kernel of the forward with its own arguments; micro-bench: the
# attention kernels) -- all on ONE tree and ONE box.   TAG=r06z tools/gpu_r06z.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); TAG=${TAG:-r06z}
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$TAG tools/gpu_round.sh > gpurun_out/round.log 2>&1
rm -rf gpurun_out/prof
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -1; timeout 300 python tools/train_slice_bench.py 2>&1 | tail -1 ) > gpurun_out/train_step.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o ${TAG}_train -- python "$REPO/tools/train_slice_bench.py" --full --steps 4 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_train.log"
cd "$REPO"
db=$(find /tmp/prof_train -name "${TAG}_train_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_train_kernel_stats.txt > /dev/null
rm -rf gpurun_out/pmc
PMC_WHICH=window6,stripe_both,refine,warp,block16 tools/gpu_pmc.sh > gpurun_out/pmc.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc gpurun_out/${TAG}_pmc > gpurun_out/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
rm -f gpurun_out/pmc/*kernel_trace.csv gpurun_out/pmc/*agent_info.csv
tail -4 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-400 gpurun_out/bench.log | tail -2; cat gpurun_out/train_step.log; tail -5 gpurun_out/pmc_traffic.log; ls gpurun_out | head -60
