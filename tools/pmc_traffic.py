#!/usr/bin/env python
"""Condense the rocprofv3 --pmc passes of tools/gpu_pmc.sh into profiles/pmc_traffic.json (HBM traffic per launch
of the attention kernels, read by bench.py for `roofline.traffic`) and a readable counter table.

    python tools/pmc_traffic.py gpurun_out/pmc profiles/r01_pmc_final
"""
import collections
import csv
import json
import os
import sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


def main(src, dst_prefix):
    table = {}
    for p in "ABCD":
        for k, cs in load(os.path.join(src, "pass%s_counter_collection.csv" % p)).items():
            if any(tag in k for tag in ("attn", "warp", "token_linear", "mlp", "nmp_block", "seed", "cost_volume", "nms", "msda")):
                table.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in cs.items()})
    # model-level passes (bench.py, eager): the conv kernels are taken from them (mean over the layers of a forward)
    # every kernel of this library that the forward itself launches is taken from the model passes (the product's own launches, all
    # layers / call sites of a kernel averaged): they overwrite the micro-benchmark figures of the same kernel name
    ours = ("conv3x3_wino", "conv3x3_split", "conv1x1", "nmp_block16", "window_attn", "stripe_attn", "warp_corr", "mlp_chain", "seed_select",
            "cost_volume", "dpn_filter", "in_stats", "in_apply", "wta_median", "refine_epilogue", "fourier_embed", "prep_images", "bias_avgpool",
            "self_attn", "msda_fwd")
    model = {}
    for pth in ("passM1", "passM2", "passM3"):
        for k, cs in load(os.path.join(src, pth + "_counter_collection.csv")).items():
            if k.startswith(ours):
                model.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in cs.items()})
    for k, c in model.items():
        table.setdefault(k, {}).update(c)
    traffic = {"_source": os.path.basename(dst_prefix) + " (rocprofv3 --pmc passes of tools/gpu_pmc.sh, mean per dispatch)"}
    for k, c in table.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            # guide (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 -> doubled;
            # WRITE_SIZE taken as is (uncalibrated); both in KiB
            traffic[k] = {"fetch_kib": c["FETCH_SIZE"], "write_kib": c["WRITE_SIZE"],
                          "hbm_bytes": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
                # MFMA pipe utilisation: busy cycles summed over the 1024 SIMDs / (kernel cycles of one XCD x 1024);
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                traffic[k]["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024), 4)
            if c.get("SQ_LDS_IDX_ACTIVE"):
                traffic[k]["lds_bank_conflict_ratio"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    with open("profiles/pmc_traffic.json", "w") as f:
        json.dump(traffic, f, indent=1)
    with open(dst_prefix + "_counters.txt", "w") as f:
        f.write("# rocprofv3 --pmc passes A-D over tools/kernel_bench.py (KITTI B=1 shapes), mean per dispatch\n"
                "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in "
                "cycles; FETCH/WRITE_SIZE in KiB\n")
        for k, v in table.items():
            f.write(k + "\n   " + json.dumps(v) + "\n")
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
