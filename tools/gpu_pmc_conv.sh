#!/bin/bash
# SQ counter passes over the layer-1 convolution alone (tools/kernel_bench.py --which conv_one) at 2 and 8 images.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); mkdir -p gpurun_out/pmc_conv
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; bb=$2; shift; shift; ( timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc_conv" -o "$tag" --output-format csv -- python "$REPO/tools/kernel_bench.py" --iters 10 --batch $bb --which conv_one 2>&1 | tail -2 ) > "$REPO/gpurun_out/pmc_conv/$tag.log"; }
for bb in 2 8; do
run b${bb}A $bb SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run b${bb}B $bb SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run b${bb}C $bb GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC
done
cd "$REPO"
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmc_conv/*_counter_collection.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv3x3_split_kernel" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-1].split("_")[0], {k: round(sum(v) / len(v)) for k, v in d.items()}, "dispatches", max((len(v) for v in d.values()), default=0))
PY
