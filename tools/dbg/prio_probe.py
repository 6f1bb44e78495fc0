"""Does a stream priority reach the parallel branches of a replayed hipGraph?  The forward has two branches (proposal stage = the critical
path; matching heads + DPN context convs on model._side_stream).  Times graph replays with the launch / capture stream and the side stream at
different priorities.   python tools/dbg/prio_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nmrf_amd.config import get_cfg
from nmrf_amd.models import build_model
from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair
from nmrf_amd import kernels as K

print("priority range:", getattr(torch.cuda.Stream, "priority_range", lambda: "n/a")())
cfg = get_cfg(); cfg.freeze()
l, r, _ = synthetic_pair(375, 1242, seed=1000)
sample = {"img1": l[None].cuda(), "img2": r[None].cuda()}


def run(tag, main_prio, side_prio, capture_on_main=True):
    model = apply_hash_weights(build_model(cfg)[0]).eval().cuda()
    main = torch.cuda.Stream(priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    if side_prio is not None:
        model._side_stream = torch.cuda.Stream(priority=side_prio)
    with torch.no_grad(), torch.cuda.stream(main):
        for _ in range(3):
            out = model(sample)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        K.kernel_hook = None
        if capture_on_main and main_prio is not None:
            with torch.cuda.graph(g, stream=main):
                out = model(sample)
        else:
            with torch.cuda.graph(g):
                out = model(sample)
        res = []
        for rep in range(3):
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 50 * 1e3)
        # eager for comparison
        t0 = time.perf_counter()
        for _ in range(20):
            model(sample)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 20 * 1e3
    print("%-70s graph replay %.3f / %.3f / %.3f ms   eager %.3f ms" % (tag, *res, eager), flush=True)


for rep in range(2):
    run("default streams (as the bench)", None, None)
    run("main high (-1), side default", -1, None)
    run("main high (-1), side low (0) explicit", -1, 0)
    run("main default, side high (-1)  [control: the wrong way round]", None, -1)
    run("main high, graph captured on the default capture stream", -1, None, capture_on_main=False)
os.environ["NMRF_OVERLAP"] = "0"
run("no side stream at all (NMRF_OVERLAP=0)", None, None)
