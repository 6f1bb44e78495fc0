#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprof kernel stats.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -20 ) > gpurun_out/smoke.log
( timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -20 ) > gpurun_out/bench.log
( timeout 600 python bench.py --steps 10 --warmup 3 --miopen-find --no-cpu-baseline 2>&1 | tail -5 ) > gpurun_out/bench_find.log
# (the find run above also leaves MIOpen's user find-db populated: on a box without it the rocprofv3 run below was seen
#  picking naive_conv / im2col solvers for some backbone convs)
cd /tmp && export TMPDIR=/tmp
( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG:-r01} -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-graph 2>&1 | tail -5 ) > "$REPO/gpurun_out/rocprof.log"
cd "$REPO"
find gpurun_out/prof -name "*stats*" | head; ls -la gpurun_out
tail -30 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log gpurun_out/bench.log gpurun_out/bench_find.log
