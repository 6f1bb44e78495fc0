#!/usr/bin/env python
"""Throughput bench of the NMRF-Stereo inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one NMRF.forward (backbone + hot path) over one batch of synthetic stereo pairs already
resident in HBM (BASELINE.json configs[1]: KITTI 1242x375, batch 1 per GPU, CNN backbone, 5/5/5
layers, fp32).  Prints ONE JSON line on rank 0: whole-job stereo pairs/s + `roofline` of the
hot-path kernel with the largest time per forward, timed live with HIP events on
the launching stream, + `other_kernels` (the next ones, incl. the N2 conv band and the HBM-bound
kernels with their GB/s) + `cpu_baseline` (the CPU oracle = a port of the reference path, timed on
the host cores, rank 0, N=1 only).  Other BASELINE configs: --height 540 --width 960 --batch 32
(config 3), --batch 8 (config 4's per-GPU shard), --backbone swin --height 1000 --width 1500
--max-disp 256 (config 5), --infer-layers 4 (the "4 inference iters" reading).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
FP16_MFMA_PEAK_TFLOPS = 2500.0       # same guide: dense fp16 / bf16 MFMA (v_mfma_f32_32x32x16_f16)
SPLIT_MFMA_PEAK_TFLOPS = FP16_MFMA_PEAK_TFLOPS / 3     # split-operand kernels execute 3 fp16 MFMAs per algorithmic product
HBM_PEAK_GBS = 8000.0                # same guide: HBM3E ~8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="stereo pairs per GPU per step")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--infer-layers", type=int, default=5, help="NMP.NUM_INFER_LAYERS (reference default 5)")
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "swin"], help="swin = configs/sceneflow_swint.yaml keys")
    ap.add_argument("--max-disp", type=int, default=320, help="DPN.MAX_DISP (256 for the Middlebury config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-figure", action="store_true", help="skip the end-to-end StereoStream figure")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL result gather at N>1")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find mode)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the result gather even at N=1 (smoke)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event brackets of named kernel launches (kernels.kernel_hook), recorded on the launching stream."""

    def __init__(self):
        self.pairs = {}
        self.meta = {}
        self.enabled = False

    def __call__(self, phase, name, meta=None):
        if not self.enabled:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        if phase == "begin":
            self.pairs.setdefault(name, []).append([ev, None])
            self.meta.setdefault(name, []).append(meta or {})
        else:
            self.pairs[name][-1][1] = ev

    def stats(self):
        """name -> (mean ms per launch, launches)"""
        out = {}
        for name, prs in self.pairs.items():
            ts = [a.elapsed_time(b) for a, b in prs if b is not None]
            if ts:
                out[name] = (sum(ts) / len(ts), len(ts))
        return out

    def mean_meta(self, name, key):
        vals = [m[key] for m in self.meta.get(name, []) if m.get(key) is not None]
        return sum(vals) / len(vals) if vals else None


def cpu_baseline(height, width, infer_layers, max_disp=320):
    """The CPU oracle (a plain-PyTorch port of the reference path, pinned to the reference by tests/test_oracle_golden.py) on
    the host cores, same synthetic pair, batch 1: 1 warm-up + median of 3 forwards on up to 16 threads, and one
    forward on 1 thread.  A bounded sample: ~10 s + ~25 s of CPU work at KITTI size."""
    from oracle import nmrf_oracle as O
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.utils.hashinit import hash_state_dict, synthetic_pair
    cores = os.cpu_count() or 1
    nthr = min(cores, 16)                     # the op-by-op CPU path stops scaling (and collapses) beyond ~16 threads
    torch.set_num_threads(nthr)
    cfg = get_cfg()
    cfg.NMP.NUM_INFER_LAYERS = infer_layers
    cfg.DPN.MAX_DISP = max_disp
    w = hash_state_dict(build_model(cfg)[0].state_dict())
    ocfg = O.OracleCfg(num_infer_layers=infer_layers, max_disp=max_disp)
    l, r, _ = synthetic_pair(height, width, seed=1000)

    def once():
        t0 = time.perf_counter()
        O.forward(w, ocfg, l[None], r[None])
        return time.perf_counter() - t0

    with torch.no_grad():
        once()                                                  # warm-up
        ts = sorted(once() for _ in range(3))
        dt = ts[1]
        torch.set_num_threads(1)
        dt1 = once()
        torch.set_num_threads(nthr)
    return {"value": 1.0 / dt, "unit": "stereo pairs/s", "cores": nthr, "kind": "port",
            "value_1thread": 1.0 / dt1,
            "sample": "one %dx%d pair, batch 1 (oracle/nmrf_oracle.py, torch CPU fp32): 1 warm-up + median of 3 forwards on %d "
                      "threads = %.2f s; one forward on 1 thread = %.2f s; host has %d cores"
                      % (width, height, nthr, dt, dt1, cores)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    import torch.distributed as dist
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from nmrf_amd import kernels as K
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.parallel import gather_disparity
    from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair

    cfg = get_cfg()
    cfg.NMP.NUM_INFER_LAYERS = args.infer_layers
    cfg.DPN.MAX_DISP = args.max_disp
    if args.backbone == "swin":                                   # configs/sceneflow_swint.yaml
        cfg.merge_from_list(["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32,
                             "BACKBONE.COMPAT", False])
    cfg.freeze()
    model = apply_hash_weights(build_model(cfg)[0]).eval().to(dev)
    b = args.batch
    pairs = [synthetic_pair(args.height, args.width, seed=1000 + rank * b + i)[:2] for i in range(b)]
    sample = {"img1": torch.stack([p[0] for p in pairs]).to(dev), "img2": torch.stack([p[1] for p in pairs]).to(dev)}

    timer = KernelTimer()
    K.kernel_hook = timer

    def forward():
        return model(sample)["disp"]

    def gather(disp):
        if use_dist and not args.no_gather:
            if world == 1:                                    # --force-dist smoke: the collective on a 1-rank group
                g = torch.empty_like(disp)
                dist.all_gather_into_tensor(g, disp.contiguous())
                return g
            return gather_disparity(disp)
        return disp

    def step():
        return gather(forward())

    graph = None
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            step()
        torch.cuda.synchronize()
        if not args.no_graph:
            try:                                          # the forward is captured; the RCCL gather stays an eager launch
                K.kernel_hook = None                      # events cannot be recorded/queried inside a capture
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = forward()
                graph.replay()
                torch.cuda.synchronize()
            except Exception as e:                        # capture unsupported -> eager launches
                print("[bench] hipGraph capture failed (%s); running eager" % str(e).splitlines()[0], file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
            K.kernel_hook = timer

        if graph is not None:
            def run():
                graph.replay()
                return gather(static_out)
        else:
            run = step
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())

        # dominant hand-written kernel, timed live with HIP events on its launching stream (eager launches,
        # same inputs, right after the timed region so clocks/caches are in the same state)
        # The side stream that overlaps the conv heads with the proposal stage is switched off for this pass (and for
        # the rocprofv3 run of tools/gpu_round.sh): a kernel sharing the chip with a MIOpen conv reports the conv's
        # duration, not its own (stripe launches of 0.42 ms measured as 4.7 ms at batch 8).
        prev_overlap = os.environ.get("NMRF_OVERLAP")
        os.environ["NMRF_OVERLAP"] = "0"
        timer.enabled = True
        n_timed_fwd = max(3, min(args.steps, 10))
        for _ in range(n_timed_fwd):
            step()
        torch.cuda.synchronize()
        timer.enabled = False
        if prev_overlap is None:
            del os.environ["NMRF_OVERLAP"]
        else:
            os.environ["NMRF_OVERLAP"] = prev_overlap
        kstats = timer.stats()

        # hot-path-only time (everything after the backbone)
        hp_ms = None
        try:
            img1, img2 = sample["img1"], sample["img2"]
            from nmrf_amd.frame_utils import InputPadder
            padder = InputPadder(img1.shape, mode="proposal", divis_by=model.divis_by)
            f1l, f2l = model.extract_feature(*padder.pad(img1, img2))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            model.hot_path(f1l, f2l, img1.shape[-2:])
            e0.record()
            for _ in range(5):
                model.hot_path(f1l, f2l, img1.shape[-2:])
            e1.record()
            torch.cuda.synchronize()
            hp_ms = e0.elapsed_time(e1) / 5
        except Exception:
            pass

        # end-to-end figure with fresh inputs per step: the double-buffered driver (nmrf_amd/driver.py) fed from HOST
        # memory -- H2D of the next batch and D2H of the previous one overlap the compute, eager launches (no hipGraph).  Reported
        # next to `value`, never as `value`.
        stream_rec = None
        if world == 1 and not args.no_stream_figure:
            try:
                from nmrf_amd.driver import StereoStream
                n_pairs = max(8, min(64, 4 * args.steps)) * 1
                host_pairs = [(i,) + tuple(t.cpu() for t in pairs[i % len(pairs)]) for i in range(n_pairs)]
                drv = StereoStream(model, dev, batch=b)
                list(drv.run(iter(host_pairs[:2 * b])))                        # warm-up
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n_done = sum(1 for _ in drv.run(iter(host_pairs)))
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t1
                stream_rec = {"value": round(n_done / dt_s, 2), "unit": "stereo pairs/s", "pairs": n_done, "batch": b,
                              "note": "nmrf_amd.driver.StereoStream: fresh host inputs per batch, pinned staging, H2D / D2H on a copy "
                                      "stream overlapped with compute, eager launches; `value` above is compute-only (hipGraph replay on "
                                      "resident inputs)"}
            except Exception as e:
                stream_rec = {"error": repr(e)}

    pairs_total = world * b * args.steps
    value = pairs_total / elapsed
    n = cfg.DPN.NUM_PROPOSALS
    # per-kernel records: ALGORITHMIC work per launch (SURVEY 8(d) formulas, recorded by the wrappers in nmrf_amd/kernels.py)
    # / mean launch time measured above.  MFMA-bound kernels are priced against the peak of the pipe they run on -- fp32 MFMA
    # (157.3 TFLOP/s) or, for the split-operand kernels, the fp16 MFMA peak / 3 (833 TFLOP/s of algorithmic products) --
    # HBM-bound ones against 8 TB/s.  `traffic` / `mfma_busy` / `lds_bank_conflict_ratio` come from separate rocprofv3 --pmc passes (never collected
    # in this run): profiles/pmc_traffic.json, whose "_source" names the round and run they belong to; they are reported only
    # for the workload those passes were taken on (KITTI, batch 1).
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        pass
    pmc_ok = b == 1 and args.height == 375 and args.width == 1242 and args.backbone == "resnet"

    def pmc_rec(names):
        for nm in names or []:
            if nm in pmc:
                return pmc[nm]
            for key, val in pmc.items():              # (a name ending in "," or "<" is a prefix: trailing template arguments vary)
                if nm[-1:] in ",<" and key.startswith(nm) and isinstance(val, dict):
                    return val
        return {}

    roof, others = None, []
    recs = []
    for k, (ms, cnt) in kstats.items():
        meta = timer.meta[k][0]
        bound = meta.get("bound", "mfma")
        flops, nbytes = timer.mean_meta(k, "flops"), timer.mean_meta(k, "bytes")
        if bound == "mfma":
            ach, unit = flops / (ms * 1e-3) / 1e12, "TFLOP/s"
            peak = round(SPLIT_MFMA_PEAK_TFLOPS, 1) if meta.get("split") else FP32_MFMA_PEAK_TFLOPS
        else:
            ach, peak, unit = nbytes / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        pr = pmc_rec(meta.get("pmc")) if pmc_ok else {}
        rec = {"bound": bound, "kernel": meta.get("label", k), "row": meta.get("row"), "achieved": round(ach, 3), "peak": peak,
               "unit": unit, "frac": round(ach / peak, 4), "traffic": pr.get("hbm_bytes"),
               "launch_ms": round(ms, 4), "launches_timed": cnt, "ms_per_forward": round(ms * cnt / n_timed_fwd, 4),
               "flop_per_launch": flops, "bytes_per_launch": nbytes}
        if bound == "mfma":
            rec["pipe"] = ("fp16 MFMA, split fp32 operands: peak = 2500 TFLOP/s / 3 products (csrc/split_mfma.h)" if meta.get("split")
                           else "fp32 MFMA")
        if pr:
            rec["traffic_source"] = "profiles/pmc_traffic.json: " + str(pmc.get("_source", "?"))
            for key in ("mfma_busy", "lds_bank_conflict_ratio"):
                if key in pr:
                    rec[key] = pr[key]
        if timer.mean_meta(k, "direct_flops"):
            df = timer.mean_meta(k, "direct_flops")
            rec["direct_form"] = {"flop_per_launch": df, "tflops": round(df / (ms * 1e-3) / 1e12, 2),
                                  "note": "SURVEY 8(d) counts the direct convolution; the kernel executes the Winograd F(2x2,3x3) "
                                          "form (1/2.25 of the multiplies), which `achieved` / `frac` are priced on"}
        recs.append(rec)
    hot = [r for r in recs if str(r["row"]).startswith("A")]
    if hot:
        roof = max(hot, key=lambda r: r["ms_per_forward"])
        others = sorted((r for r in recs if r is not roof), key=lambda r: -r["ms_per_forward"])[:9]

    if rank == 0:
        res = {
            "metric": "stereo pairs/sec at %dx%d" % (args.width, args.height), "value": round(value, 3), "unit": "stereo pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32" if os.environ.get("NMRF_LINEAR", "split") == "fp32" else
                      "f32 (storage, accumulation, softmax / LayerNorm / GELU, convolutions; the contractions of the per-token "
                      "linears and of the attention kernels multiply fp32 operands as fp16 hi/lo pairs on the fp16 MFMA, 3 products "
                      "per term, ~2^-22 relative, fp32 accumulate -- csrc/split_mfma.h; NMRF_LINEAR=fp32 runs the linears on "
                      "the fp32 MFMA)"),
            "data": "synthetic",
            "config": {"workload": "%s %dx%d stereo pairs, batch %d per GPU, %s backbone, D_max %d, %d/%d/%d prop/infer/refine "
                                   "layers, hash-formula weights" % (
                                       {(375, 1242): "KITTI", (540, 960): "SceneFlow", (1000, 1500): "Middlebury-H"}.get(
                                           (args.height, args.width), "synthetic"), args.width, args.height, b,
                                       "CNN" if args.backbone == "resnet" else "Swin-T + deformable neck (HIP MSDA)",
                                       args.max_disp, cfg.NMP.NUM_PROP_LAYERS, cfg.NMP.NUM_INFER_LAYERS,
                                       cfg.NMP.NUM_REFINE_LAYERS),
                       "global_batch": world * b, "parallelism": "batch-shard x%d" % world,
                       "launch": "hipGraph" if graph is not None else "eager",
                       "result_gather": bool(use_dist and not args.no_gather)},
            "hot_path_ms": None if hp_ms is None else round(hp_ms, 3),
            "roofline": roof,
            "other_kernels": others,
        }
        if stream_rec is not None:
            res["stream_end_to_end"] = stream_rec
        if world == 1 and not args.no_cpu_baseline and args.backbone == "resnet":      # the oracle restates the CNN configuration
            try:
                res["cpu_baseline"] = cpu_baseline(args.height, args.width, args.infer_layers, args.max_disp)
            except Exception as e:
                res["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
