#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1200 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "k384 or z312 or crafted or small_gradients or backward_gemm or reference_prob or two_kernel_path" 2>&1 | tail -60 ) > gpurun_out/r06d_pytest.log
cat gpurun_out/r06d_pytest.log
