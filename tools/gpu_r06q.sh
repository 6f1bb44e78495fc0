#!/bin/bash
# round 6, call Q: one-launch tree reduction -- gradient tests + the training step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "train or backward or gradient or bwd or dropout" 2>&1 | tail -25 ) > gpurun_out/r06q_pytest.log
tail -12 gpurun_out/r06q_pytest.log | cut -c1-250
TAG=r06q bash tools/gpu_r06p.sh 2>&1 | head -14 | cut -c1-160
