#!/bin/bash
# round 5, call D: is the block kernels' weight stream (every CU reads the SAME 16 KB stages from L2) the bound?  Replicated streams.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 300 python tools/mlp_variants_bench.py --iters 30 --tokens 29952,958464 2>&1 | grep -v amdgpu.ids | tail -26 ) > gpurun_out/r05d_stream_copies.txt
cat gpurun_out/r05d_stream_copies.txt
