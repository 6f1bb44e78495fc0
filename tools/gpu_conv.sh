#!/bin/bash
# Short gpurun call for the conv kernels: their parity tests, the per-layer micro-bench, one bench line.   TAG=r02m tools/gpu_conv.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
TAG=${TAG:-r02}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "${PYTEST_K:-conv3x3 or conv1x1}" 2>&1 | tail -60 ) > gpurun_out/pytest_conv.log
( timeout 600 python tools/kernel_bench.py --iters 20 --which conv 2>&1 | grep -v stamp | tail -20 ) > gpurun_out/kernel_bench_conv.log
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_conv.log
if [ -n "$WITH_E2E" ]; then
  ( timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -rf -p no:cacheprovider -x 2>&1 | tail -40 ) > gpurun_out/pytest_model.log
fi
cat gpurun_out/pytest_conv.log; cat gpurun_out/kernel_bench_conv.log; cut -c1-1200 gpurun_out/bench_conv.log; [ -n "$WITH_E2E" ] && cat gpurun_out/pytest_model.log
