#!/bin/bash
# round 6, call H: same-box A/B of the one-transcendental GELU (A = Abramowitz-Stegun form of rounds 2-5, B = tree: packed pairs;
# then A = the new form un-packed, B = tree), then the whole GPU suite on the tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
A=nmrf_amd/lib/ab_gelu_as/libnmrf_hip.so B=nmrf_amd/lib/libnmrf_hip.so TAG=r06h1 tools/gpu_ab.sh > gpurun_out/r06h_gelu_as_vs_tree.log 2>&1
A=nmrf_amd/lib/ab_gelu_sc/libnmrf_hip.so B=nmrf_amd/lib/libnmrf_hip.so TAG=r06h2 tools/gpu_ab.sh > gpurun_out/r06h_gelu_scalar_vs_tree.log 2>&1
grep -h "nmp_block16\|mlp_chain\|^A:\|^B:\|total kernel" gpurun_out/r06h_gelu_as_vs_tree.log gpurun_out/r06h_gelu_scalar_vs_tree.log | cut -c1-150
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -60 ) > gpurun_out/r06h_pytest.log
grep -h "WTA decisions" gpurun_out/r06h_pytest.log | cut -c1-260
tail -8 gpurun_out/r06h_pytest.log
