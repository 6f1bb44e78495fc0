#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( echo "== plain"; AMD_LOG_LEVEL=1 timeout 300 python tools/dbg/stripe_launch.py 2>&1 | grep -v amdgpu.ids | tail -25; echo "== with both libraries preloaded (conftest's stamp check)"; timeout 300 python tools/dbg/stripe_launch.py preload 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r06o_stripe_launch.log
cat gpurun_out/r06o_stripe_launch.log | cut -c1-300
