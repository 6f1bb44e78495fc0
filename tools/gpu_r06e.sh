#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > gpurun_out/r06e_stripe.log
for rep in 1 2; do for v in ab/libnmrf_hip_debug.so libnmrf_hip_debug.so; do
  echo "== $v" >> gpurun_out/r06e_stripe.log
  ( timeout 600 python tools/kernel_bench.py --iters 50 --which stripe_both --lib nmrf_amd/lib/$v 2>&1 | tail -1 ) >> gpurun_out/r06e_stripe.log
  ( timeout 600 python tools/kernel_bench.py --iters 20 --batch 8 --which stripe_both --lib nmrf_amd/lib/$v 2>&1 | tail -1 ) >> gpurun_out/r06e_stripe.log
done; done
cat gpurun_out/r06e_stripe.log
