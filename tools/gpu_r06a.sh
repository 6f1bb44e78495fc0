#!/bin/bash
# round 6, call A: the persistent window kernel -- parity tests + same-box A/B against the two-windows-per-block kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "window" 2>&1 | tail -60 ) > gpurun_out/r06a_pytest_window.log
( timeout 600 python tools/kernel_bench.py --iters 50 --which window6 2>&1 | tail -12 ) > gpurun_out/r06a_window6_ab.log
( timeout 600 python tools/kernel_bench.py --iters 20 --batch 8 --which window6 2>&1 | tail -4 ) >> gpurun_out/r06a_window6_ab.log
cat gpurun_out/r06a_pytest_window.log gpurun_out/r06a_window6_ab.log
