#!/bin/bash
# round 6, call N: LePE of the stripe kernels on the matrix pipe -- tests, then same-call A/B (ab/ = stripe_attn.hip of HEAD)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q --tb=short -rf -p no:cacheprovider -k "stripe" 2>&1 | tail -30 ) > gpurun_out/r06n_pytest.log
tail -30 gpurun_out/r06n_pytest.log | cut -c1-300
: > gpurun_out/r06n_stripe_ab.log
for rep in 1 2; do for v in ab/libnmrf_hip_debug.so libnmrf_hip_debug.so; do
  echo "== $v" >> gpurun_out/r06n_stripe_ab.log
  ( timeout 600 python tools/kernel_bench.py --iters 50 --which stripe_both --lib nmrf_amd/lib/$v 2>&1 | tail -1 ) >> gpurun_out/r06n_stripe_ab.log
  ( timeout 600 python tools/kernel_bench.py --iters 20 --batch 8 --which stripe_both --lib nmrf_amd/lib/$v 2>&1 | tail -1 ) >> gpurun_out/r06n_stripe_ab.log
done; done
cat gpurun_out/r06n_stripe_ab.log
