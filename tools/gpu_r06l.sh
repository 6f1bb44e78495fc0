#!/bin/bash
# round 6, call L: does the clock-record hook cost the block kernels anything?  (A = compiled out, B = tree) + a bench line with the clock
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
A=nmrf_amd/lib/ab_noclk/libnmrf_hip.so B=nmrf_amd/lib/libnmrf_hip.so TAG=r06l tools/gpu_ab.sh > gpurun_out/r06l_clk_hook_ab.log 2>&1
grep -h "nmp_block16\|^A:\|^B:\|total kernel" gpurun_out/r06l_clk_hook_ab.log | cut -c1-150
( timeout 900 python bench.py --no-cpu-baseline --no-stream-figure 2>&1 | tail -1 ) > gpurun_out/r06l_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06l_bench.json").read())
print(d["value"], d["ms_per_step"], d["sustained_clock_ghz"], {k: d["roofline"].get(k) for k in ("frac", "launch_ms", "sustained_clock_ghz", "frac_at_sustained_clock")})
PY
