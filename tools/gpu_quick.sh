#!/bin/bash
# quick GPU check: selected tests + bench (+ kernel micro-bench).  PYTEST_K='range or warp' KB=block tools/gpu_quick.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "${PYTEST_K:-range}" 2>&1 | tail -40 ) > gpurun_out/pytest_quick.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -3 ) > gpurun_out/bench_quick.log
tail -30 gpurun_out/pytest_quick.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_quick.log').read().splitlines() if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d.get(k) for k in ('value','ms_per_step','hot_path_ms')})
    print({k:(d.get(k) or {}).get('value') for k in ('stream_end_to_end','stream_end_to_end_b8')})
    for r in [d['roofline']]+d['other_kernels']:
        print("%-60s %-5s frac %.3f  %.1f us x %.1f ms/fwd" % (r['kernel'][:60], r['bound'], r['frac'], r['launch_ms']*1e3, r['ms_per_forward']))
else:
    print(open('gpurun_out/bench_quick.log').read()[-2000:])
PY
if [ -n "$KB" ]; then ( timeout 600 python tools/kernel_bench.py --iters 20 --which $KB 2>&1 | grep -v stamp | tail -${KBTAIL:-60} ) > gpurun_out/kernel_bench_quick.log; cat gpurun_out/kernel_bench_quick.log; fi
