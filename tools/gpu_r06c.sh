#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "window" 2>&1 | tail -30 ) > gpurun_out/r06c_pytest_window.log
( timeout 600 python tools/kernel_bench.py --iters 30 --which window6,window6_census,window6_stamps 2>&1 | tail -50 ) > gpurun_out/r06c_window6.log
( timeout 600 python tools/kernel_bench.py --iters 10 --batch 8 --which window6,window6_census,window6_stamps 2>&1 | tail -50 ) >> gpurun_out/r06c_window6.log
cat gpurun_out/r06c_pytest_window.log gpurun_out/r06c_window6.log
