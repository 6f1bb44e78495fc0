#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "hot_path_from_reference_features or stages_from_reference_inputs or individual_layers" 2>&1 | tail -80 ) > gpurun_out/pytest_r05j.log
cat gpurun_out/pytest_r05j.log
