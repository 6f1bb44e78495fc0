#!/bin/bash
# round 5, call C: packed-fp32 VALU beside MFMAs -- the block / chain kernels and the b32 micro-kernel built with and without
# v_pk_* (tools/build_ab_nopk.sh), same box: micro-bench rows, eager kernel traces A/B, graph-replay bench lines A,B,A,B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 300 python tools/mlp_variants_bench.py --iters 30 --tokens 29952,958464 2>&1 | grep -v amdgpu.ids | tail -14 ) > gpurun_out/r05c_mlp_variants.txt
( timeout 300 python tools/mlp_variants_bench.py --iters 30 --tokens 29952,958464 --lib nmrf_amd/lib/ab_nopk/libnmrf_hip.so --so tools/_ab/mlp_b32_nopk.so 2>&1 | grep -v amdgpu.ids | tail -14 ) >> gpurun_out/r05c_mlp_variants.txt
cat gpurun_out/r05c_mlp_variants.txt
( timeout 300 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "nmp_block or mlp_chain or heads_wta or refine_head" 2>&1 | tail -4 ) > gpurun_out/pytest_nopk_main.log
tail -3 gpurun_out/pytest_nopk_main.log
A=nmrf_amd/lib/libnmrf_hip.so B=nmrf_amd/lib/ab_nopk/libnmrf_hip.so TAG=r05c tools/gpu_ab.sh 2>&1 | tail -45
