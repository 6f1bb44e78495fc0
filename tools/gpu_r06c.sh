#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python tools/kernel_bench.py --iters 30 --which window6_census,window6_stamps 2>&1 | tail -50 ) > gpurun_out/r06c_window6.log
( timeout 600 python tools/kernel_bench.py --iters 10 --batch 8 --which window6_census,window6_stamps 2>&1 | tail -50 ) >> gpurun_out/r06c_window6.log
cat gpurun_out/r06c_window6.log
