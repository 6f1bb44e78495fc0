#!/bin/bash
# round 6, call B: census of the persistent window kernel (timing ablations) + SQ counters of both window kernels at batch 8
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); mkdir -p gpurun_out/pmc6
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 600 python tools/kernel_bench.py --iters 30 --which window6_census 2>&1 | tail -12 ) > gpurun_out/r06b_window6_census.log
( timeout 600 python tools/kernel_bench.py --iters 10 --batch 8 --which window6_census 2>&1 | tail -12 ) >> gpurun_out/r06b_window6_census.log
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; ( timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc6" -o "$tag" --output-format csv -- python "$REPO/tools/kernel_bench.py" --iters 2 --batch 8 --which window6 2>&1 | tail -3 ) > "$REPO/gpurun_out/pmc6/$tag.log"; }
run passA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run passB SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run passC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC
cd "$REPO"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc6/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "window_attn" not in k: continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    print("  ", {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "dispatches", max(len(v) for v in d.values()))
PY
cat gpurun_out/r06b_window6_census.log
