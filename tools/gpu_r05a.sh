#!/bin/bash
# round 5, call A: full GPU parity suite on the new gates / fixtures + the bench with the RCCL 1-rank group and the config-4 leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/e2e_stats.jsonl gpurun_out/chain_stats.jsonl
( timeout 1500 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider 2>&1 | tail -150 ) > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
( timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --config4 --no-cpu-baseline 2>&1 | tail -3 ) > gpurun_out/bench_config4.log
tail -90 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_config4.log').read().splitlines() if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d.get(k) for k in ('value','ms_per_step','hot_path_ms','config4')})
    print(d['config'].get('gather'), d['config'].get('numa_pinning_rank0'))
    print(d.get('stream_end_to_end'))
else:
    print(open('gpurun_out/bench_config4.log').read()[-3000:])
PY
