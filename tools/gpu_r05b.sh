#!/bin/bash
# round 5, call B: MLP-phase decompositions + the two ADVICE tests + the bench with the RCCL 1-rank group / config-4 leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python tools/mlp_variants_bench.py --iters 30 2>&1 | tail -40 ) > gpurun_out/r05b_mlp_variants.txt
cat gpurun_out/r05b_mlp_variants.txt
( timeout 600 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider -k "table_magnitude or non_finite or kv16_rows" 2>&1 | tail -15 ) > gpurun_out/pytest_quick.log
tail -8 gpurun_out/pytest_quick.log
( timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --config4 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 ) > gpurun_out/bench_config4.log
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_config4.log').read().splitlines() if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d.get(k) for k in ('value','ms_per_step','hot_path_ms','config4')})
    print(d['config'].get('gather'), d['config'].get('numa_pinning_rank0'))
    print(d.get('stream_end_to_end'))
else:
    print("no bench line")
PY
