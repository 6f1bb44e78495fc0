#!/bin/bash
# One gpurun call: clean rocprofv3 kernel stats of the default bench (no stream figure, side stream off) + the PMC passes.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); TAG=${TAG:-r03}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG} -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-stream-figure --no-graph 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof.log"
cd "$REPO"
db=$(find gpurun_out/prof -name "${TAG}_results.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${TAG}_kernel_stats.txt > /dev/null
PMC_WHICH=window,stripe,refine,warp,block tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc gpurun_out/${TAG}_pmc > gpurun_out/pmc_traffic_print.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
head -50 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-160; tail -5 gpurun_out/pmc_run.log
