#!/bin/bash
# round 6, call M: stripe census (where a horizontal-stripe block's cycles go) + the final window_attn6 same-call A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python tools/kernel_bench.py --iters 30 --which stripe_census,stripe,stripe_both 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06m_stripe_census.log
( timeout 600 python tools/kernel_bench.py --iters 30 --which window6 2>&1 | grep -v amdgpu.ids; timeout 600 python tools/kernel_bench.py --iters 10 --batch 8 --which window6 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06m_window6_ab.log
cat gpurun_out/r06m_stripe_census.log gpurun_out/r06m_window6_ab.log | cut -c1-250
