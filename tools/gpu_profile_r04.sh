#!/bin/bash
# Round-4 profile refresh in ONE gpurun call: kernel stats of the default bench and of config 3 (SceneFlow batch 32),
# the PMC passes of tools/gpu_pmc.sh for the final-tree kernels, and the block kernel's SQ counters at KITTI batch 1 AND at batch 32
# tokens (the ceiling model of DESIGN section 10.1).   TAG=r04 tools/gpu_profile_r04.sh
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
REPO=$(pwd); TAG=${TAG:-r04}
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
( NMRF_OVERLAP=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG} -- python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-stream-figure --no-graph 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof.log"
( NMRF_OVERLAP=0 timeout 900 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o ${TAG}_sf32 -- python "$REPO/bench.py" --steps 2 --warmup 1 --batch 32 --height 540 --width 960 --no-cpu-baseline --no-stream-figure --no-graph 2>&1 | tail -3 ) > "$REPO/gpurun_out/rocprof_sf32.log"
cd "$REPO"
for t in ${TAG} ${TAG}_sf32; do
  db=$(find gpurun_out/prof -name "${t}_results.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" gpurun_out/${t}_kernel_stats.txt > /dev/null
done
PMC_WHICH=window,stripe,refine,warp,block tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc gpurun_out/${TAG}_pmc > gpurun_out/pmc_traffic_print.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
# block kernel at batch-32 token count: SQ passes only
cd /tmp
runb() { tag=$1; shift; ( timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc" -o "$tag" --output-format csv -- python "$REPO/tools/kernel_bench.py" --iters 2 --batch 32 --which block16 2>&1 | tail -3 ) > "$REPO/gpurun_out/pmc/$tag.log"; }
runb b32A SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
runb b32B SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
runb b32C GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC
runb1() { tag=$1; shift; ( timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$REPO/gpurun_out/pmc" -o "$tag" --output-format csv -- python "$REPO/tools/kernel_bench.py" --iters 3 --batch 1 --which block16 2>&1 | tail -3 ) > "$REPO/gpurun_out/pmc/$tag.log"; }
runb1 b01C GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_MISC
cd "$REPO"
python - <<'PY'
import csv, collections, json, os
out = {}
for tag in ("passA", "passB", "b01C", "b32A", "b32B", "b32C"):
    p = "gpurun_out/pmc/%s_counter_collection.csv" % tag
    if not os.path.exists(p):
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "nmp_block16_kernel<true, 5" in k:
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in d.items():
        out.setdefault(("B=32 tokens " if tag.startswith("b32") else "B=1 tokens ") + k, {}).update({c: round(sum(v) / len(v)) for c, v in cs.items()})
json.dump(out, open("gpurun_out/r04_block_sq_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
ls gpurun_out/pmc | head -40; head -30 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-150; head -24 gpurun_out/${TAG}_sf32_kernel_stats.txt | cut -c1-150; tail -3 gpurun_out/pmc_run.log
