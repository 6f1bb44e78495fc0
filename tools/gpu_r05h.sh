#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "${PYTEST_K:-backward or autograd or training or train_steps}" 2>&1 | tail -40 ) > gpurun_out/pytest_n4.log
cat gpurun_out/pytest_n4.log | cut -c1-400
