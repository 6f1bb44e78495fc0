#!/bin/bash
# round 6, call K: the whole GPU suite + a driver-style bench line on the tree (one-transcendental GELU, stock split, clock sampler)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/r06k_pytest.log
grep -h "WTA decisions" gpurun_out/r06k_pytest.log | cut -c1-260
tail -5 gpurun_out/r06k_pytest.log
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r06k_bench.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06k_bench.json").read())
print(d["value"], d["ms_per_step"], d["sustained_clock_ghz"], {k: d["roofline"].get(k) for k in ("frac", "launch_ms", "sustained_clock_ghz", "frac_at_sustained_clock")}, d["cpu_baseline"]["value"])
PY
