#!/bin/bash
# round 5, call E: N4 first slice -- backward pieces, autograd Functions, the model-level gradient test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "backward or autograd or training or train_steps" ${PYTEST_EXTRA} 2>&1 | tail -60 ) > gpurun_out/pytest_n4.log
cat gpurun_out/pytest_n4.log | cut -c1-400
