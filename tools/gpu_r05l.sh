#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "${KEXPR:-backward or train_steps or autograd}" 2>&1 | tail -40 ) > gpurun_out/pytest_r05l.log
cat gpurun_out/pytest_r05l.log
( timeout 300 python tools/train_slice_bench.py --full 2>&1 | tail -1;  timeout 300 python tools/train_slice_bench.py 2>&1 | tail -1 ) > gpurun_out/train_full.log
cat gpurun_out/train_full.log
