"""MSDA forward inside the Swin-T configuration's forward, tiled (product) vs untiled (tools variant 4), same process, same inputs:
HIP-event time of each of the neck's four calls.   python tools/dbg/msda_model_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nmrf_amd._lib as L
L.LIB_PATH = L.DEBUG_LIB_PATH                       # the tools library is a superset of the product library (same ABI)
from nmrf_amd.config import get_cfg
from nmrf_amd.models import build_model
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair
from nmrf_amd import kernels as K
lib = L.load()
cfg = get_cfg()
cfg.merge_from_list(["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32, "BACKBONE.COMPAT", False, "DPN.MAX_DISP", 256])
cfg.freeze()
model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
l, r, _ = synthetic_pair(1000, 1500, seed=1000)
sample = {"img1": l[None].cuda(), "img2": r[None].cuda()}
ev = []
def hook(phase, name, meta=None):
    if name != "msda_forward":
        return
    e = torch.cuda.Event(enable_timing=True); e.record()
    if phase == "begin": ev.append([e, None])
    else: ev[-1][1] = e
with torch.no_grad():
    for _ in range(2):
        model(sample)
    for rep in range(3):
        for var, tag in ((0, "tiled"), (4, "untiled")):
            lib.nmrf_debug_msda_variant(var)
            ev.clear()
            K.kernel_hook = hook
            for _ in range(5):
                model(sample)
            torch.cuda.synchronize()
            K.kernel_hook = None
            ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
            per = [sum(ts[i::4]) / len(ts[i::4]) for i in range(4)]
            print("%-8s per call (levels 1/4 .. 1/32): %s  sum %.1f us" % (tag, " ".join("%6.1f" % t for t in per), sum(per)), flush=True)
lib.nmrf_debug_msda_variant(0)
