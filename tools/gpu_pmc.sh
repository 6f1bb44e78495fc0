#!/bin/bash
# PMC counter passes over the attention micro-bench (separate runs, kernel-trace only -- see task notes).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); mkdir -p gpurun_out/pmc
[ -z "$PMC_SKIP_KERNELS" ] && python tools/kernel_bench.py --iters 20 --which ${PMC_WHICH:-window,stripe,refine,warp,block} > gpurun_out/pmc/kernel_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$REPO/gpurun_out/pmc/counters.txt" 2>&1
run() { tag=$1; shift; ( timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc" -o "$tag" --output-format csv -- python "$REPO/tools/kernel_bench.py" --iters 3 --which ${PMC_WHICH:-window,stripe,refine,warp,block} 2>&1 | tail -3 ) > "$REPO/gpurun_out/pmc/$tag.log"; }
if [ -z "$PMC_SKIP_KERNELS" ]; then
run passA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run passB SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run passC FETCH_SIZE GRBM_GUI_ACTIVE
run passD WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
fi
# the conv kernel runs 9-12 different layers per forward: its traffic is collected on the model itself (eager, 2 forwards)
runm() { tag=$1; shift; ( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc" -o "$tag" --output-format csv -- python "$REPO/bench.py" --steps 2 --warmup 1 --no-graph --no-clock-sample --no-cpu-baseline --no-stream-figure 2>&1 | tail -2 ) > "$REPO/gpurun_out/pmc/$tag.log"; }
if [ -z "$PMC_SKIP_MODEL" ]; then
runm passM1 FETCH_SIZE GRBM_GUI_ACTIVE
runm passM2 WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
runm passM3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES
fi
cd "$REPO"; ls -la gpurun_out/pmc; cat gpurun_out/pmc/kernel_bench.log; tail -2 gpurun_out/pmc/pass*.log; grep -c . gpurun_out/pmc/counters.txt
