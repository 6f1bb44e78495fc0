#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for l in nmrf_amd/lib/libnmrf_hip.so nmrf_amd/lib/ab_gelu_as/libnmrf_hip.so nmrf_amd/lib/ab_gelu_sc/libnmrf_hip.so; do
  timeout 300 python tools/dbg/gelu_check.py $l 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06i_gelu_check.log
cat gpurun_out/r06i_gelu_check.log
( timeout 600 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "mlp_chain or nmp_block" 2>&1 | tail -80 ) > gpurun_out/r06i_pytest.log
tail -80 gpurun_out/r06i_pytest.log | cut -c1-300
