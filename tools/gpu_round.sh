#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (default = BASELINE config 2) + the other BASELINE configs, rocprof kernel stats.
# Outputs under gpurun_out/.   TAG=r02a tools/gpu_round.sh [quick]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd)
TAG=${TAG:-r05}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/e2e_stats.jsonl
( timeout 2400 python -m pytest tests -m gpu -q --tb=short -rf -p no:cacheprovider ${PYTEST_ARGS} 2>&1 | tail -200 ) > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -20 ) > gpurun_out/smoke.log
( timeout 900 python bench.py 2>&1 | tail -5 ) > gpurun_out/bench.log      # the driver's command: defaults (100 timed steps)
if [ -z "$LEAN" ]; then      # LEAN=1: tests, smoke, bench and the kernel trace only
( NMRF_LINEAR=fp32 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_fp32_linears.log
( timeout 600 python tools/kernel_bench.py --iters 20 --which window,stripe,refine,block 2>&1 | grep -v stamp | tail -60 ) > gpurun_out/kernel_bench_block.log
fi
if [ "$1" != "quick" ]; then
  # the RCCL path on the one GPU of the box (1-rank group, the gather in the loop) + BASELINE config 4's per-GPU shard as its own record
  ( timeout 600 python bench.py --steps 20 --warmup 5 --force-dist --config4 --no-cpu-baseline --no-stream-figure 2>&1 | grep '^{' | tail -1 ) > gpurun_out/bench_config4.log
  # the other BASELINE configs: bench lines kept under profiles/
  ( timeout 600 python bench.py --steps 10 --warmup 3 --infer-layers 4 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_infer4.log
  ( timeout 600 python bench.py --steps 5 --warmup 2 --batch 8 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_kitti_b8.log
  ( timeout 900 python bench.py --steps 3 --warmup 2 --batch 32 --height 540 --width 960 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_sceneflow_b32.log
  ( timeout 900 python bench.py --steps 5 --warmup 2 --backbone swin --height 1000 --width 1500 --max-disp 256 --no-cpu-baseline 2>&1 | tail -2 ) > gpurun_out/bench_swin_middlebury.log
fi
cd /tmp && export TMPDIR=/tmp
( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG} -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-stream-figure --no-graph --no-clock-sample 2>&1 | tail -5 ) > "$REPO/gpurun_out/rocprof.log"
if [ "$1" != "quick" ]; then
  ( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG}_swin -- python "$REPO/bench.py" --steps 2 --warmup 1 --backbone swin --height 1000 --width 1500 --max-disp 256 --no-cpu-baseline --no-stream-figure --no-graph --no-clock-sample 2>&1 | tail -5 ) > "$REPO/gpurun_out/rocprof_swin.log"
fi
cd "$REPO"
for t in ${TAG} ${TAG}_swin; do
  db=$(find gpurun_out/prof -name "${t}_results.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${t}_kernel_stats.txt > /dev/null
done
ls -la gpurun_out
tail -60 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; for f in gpurun_out/bench*.log; do echo "== $f"; cut -c1-1500 $f; done
[ -z "$LEAN" ] && cat gpurun_out/kernel_bench_block.log
head -40 gpurun_out/${TAG}_kernel_stats.txt
