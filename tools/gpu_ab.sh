#!/bin/bash
# Same-box A/B of two builds of libnmrf_hip.so: eager kernel traces of bench.py with each, then graph-replay bench lines A,B,A,B.
#   A=nmrf_amd/lib/ab_main/libnmrf_hip.so B=nmrf_amd/lib/libnmrf_hip.so TAG=r04a tools/gpu_ab.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); TAG=${TAG:-r04ab}
A=${A:-nmrf_amd/lib/ab_main/libnmrf_hip.so}; B=${B:-nmrf_amd/lib/libnmrf_hip.so}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  lib=$REPO/${!v}
  ( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG}_$v -- python "$REPO/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-stream-figure --no-graph --no-clock-sample --lib $lib ${BENCH_ARGS} 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_$v.log"
done
cd "$REPO"
da=$(find gpurun_out/prof -name "${TAG}_A_results.db" | head -1); db=$(find gpurun_out/prof -name "${TAG}_B_results.db" | head -1)
python tools/ab_table.py "$da" "$db" gpurun_out/${TAG}_ab.txt A B | cut -c1-200
for i in 1 2; do for v in A B; do
  echo -n "$v: "; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-stream-figure --lib $REPO/${!v} ${BENCH_ARGS} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('hot_path_ms'))"
done; done | tee -a gpurun_out/${TAG}_ab.txt
