#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for l in nmrf_amd/lib/ab_nomix/libnmrf_hip.so nmrf_amd/lib/ab_mixnop/libnmrf_hip.so; do
  echo "== $l"; timeout 300 python tools/dbg/chain1_diag.py $l 2>&1 | grep -v amdgpu.ids | head -6
done > gpurun_out/r06j_chain1_diag2.log
cat gpurun_out/r06j_chain1_diag2.log
