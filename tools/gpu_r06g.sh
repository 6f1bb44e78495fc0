#!/bin/bash
# round 6, call G: same-box A/B of the v_fma_mix split (A = -DNMRF_NO_MIX_SPLIT build, B = tree), then the whole GPU suite on the tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
A=nmrf_amd/lib/ab_nomix/libnmrf_hip.so B=nmrf_amd/lib/libnmrf_hip.so TAG=r06g tools/gpu_ab.sh > gpurun_out/r06g_ab.log 2>&1
tail -40 gpurun_out/r06g_ab.log | cut -c1-220
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -x 2>&1 | tail -40 ) > gpurun_out/r06g_pytest.log
tail -8 gpurun_out/r06g_pytest.log
