#!/bin/bash
# One gpurun call: the end-to-end parity tests + the flip-floor table (GPU rows).  TAG=r03a tools/gpu_parity.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
TAG=${TAG:-r03}
mkdir -p gpurun_out
rm -f gpurun_out/e2e_stats.jsonl gpurun_out/chain_stats.jsonl
( timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_swin_config.py -m gpu -q --tb=short -rf -p no:cacheprovider ${PYTEST_ARGS} 2>&1 | tail -150 ) > gpurun_out/pytest_model.log
( timeout 900 python tools/flip_floor.py --kitti --out gpurun_out/${TAG}_flip_floor.md 2>&1 | tail -80 ) > gpurun_out/flip_floor.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 ) > gpurun_out/smoke.log
tail -120 gpurun_out/pytest_model.log; cat gpurun_out/flip_floor.log gpurun_out/smoke.log
