#!/bin/bash
# rocprofv3 kernel trace of the hipGraph-replay bench (the product's launch mode): gaps between dependent kernels, overlap of the
# side streams.  Output: gpurun_out/prof/<TAG>_results.db + gpurun_out/<TAG>_timeline.txt   TAG=r03i tools/gpu_graph_trace.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd)
TAG=${TAG:-graphtrace}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG} -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-stream-figure ${BENCH_ARGS} 2>&1 | tail -3 ) > "$REPO/gpurun_out/${TAG}_rocprof.log"
cd "$REPO"
db=$(find gpurun_out/prof -name "${TAG}_results.db" | head -1)
python tools/graph_timeline.py "$db" > gpurun_out/${TAG}_timeline.txt 2>&1
tail -5 gpurun_out/${TAG}_rocprof.log | cut -c1-300
head -200 gpurun_out/${TAG}_timeline.txt
