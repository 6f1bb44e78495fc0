#!/bin/bash
# round 6, call S: the tiled MSDA forward -- tests, then timing against the untiled form at the four levels of the neck
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_swin_config.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "msda or swin" 2>&1 | tail -30 ) > gpurun_out/r06s_pytest.log
tail -30 gpurun_out/r06s_pytest.log | cut -c1-300
( timeout 600 python tools/kernel_bench.py --iters 30 --which msda 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06s_msda.log
cat gpurun_out/r06s_msda.log | cut -c1-200
