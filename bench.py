#!/usr/bin/env python
"""Throughput bench of the NMRF-Stereo inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                     # N > 1: starts its own N workers (one per GPU), like
                                                                      # the reference's launcher (main.py:87-144, mp.start_processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   # same workers

Without a GPU (build container) `--gpus N` still runs: N gloo ranks on CPU with a stand-in forward -- launcher / batch split /
gather plumbing only, the JSON line says `"valid": false`.

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
BOOST_CLOCK_GHZ = 2.4                 # same guide: the engine clock its peak figures are quoted at


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="stereo pairs per GPU per step")
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--infer-layers", type=int, default=5, help="NMP.NUM_INFER_LAYERS (reference default 5)")
    ap.add_argument("--backbone", default="resnet", choices=["resnet", "swin"], help="swin = configs/sceneflow_swint.yaml keys")
    ap.add_argument("--max-disp", type=int, default=320, help="DPN.MAX_DISP (256 for the Middlebury config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream-figure", action="store_true", help="skip the end-to-end StereoStream figure")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-clock-sample", action="store_true", help="skip the 320 back-to-back launches of the dominant kernel behind "
                    "sustained_clock_ghz (kernel traces / --pmc passes of this command: they would dominate the per-kernel totals)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL result gather at N>1")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find mode)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the result gather even at N=1 (smoke)")
    ap.add_argument("--config4", action="store_true", help="also measure BASELINE config 4's per-GPU shard (batch 8) at N=1 (always on at N>1)")
    ap.add_argument("--sync-gather", action="store_true", help="gather on the compute stream inside every step (round-2 behaviour)")
    ap.add_argument("--lib", default=None, help="same-box A/B runs (tools/gpu_ab.sh): another build of libnmrf_hip.so (same ABI, checked at load)")
    args = ap.parse_args(argv)
    if args.lib:
        import nmrf_amd._lib as _L
        _L.LIB_PATH = os.path.abspath(args.lib)
    return args


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


def _set_infer_layers(cfg, n):
    """NMP.NUM_INFER_LAYERS with SOLVER.LOSS_WEIGHTS kept one-per-layer: build() asserts that, as nmrf/models/NMRF.py:434 does."""
    lw, n0 = list(cfg.SOLVER.LOSS_WEIGHTS), cfg.NMP.NUM_INFER_LAYERS
    cfg.NMP.NUM_INFER_LAYERS = n
    cfg.SOLVER.LOSS_WEIGHTS = lw[n0 - n:] if n <= n0 else [lw[0]] * (n - n0) + lw


def cpu_baseline(height, width, infer_layers, max_disp=320):
    """The CPU oracle (a plain-PyTorch port of the reference path, pinned to the reference by tests/test_oracle_golden.py) on the
    host cores, same synthetic pairs, batch 1 (both 1242x375 and 960x540, all host cores, and a 1-thread figure).
    The op-by-op CPU path stops scaling at ~16 threads, so each size is timed on 16 threads AND on every core of the host;
    `value` / `cores` is the faster of the two at the bench's own size, the other numbers ride along.  A bounded sample:
    1 warm-up + median of 5 forwards on 16 threads at the bench's own size (3 at the other size) and 1 warm-up + 2 forwards on all
    cores; one forward on 1 thread."""
    from oracle import nmrf_oracle as O
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.utils.hashinit import hash_state_dict, synthetic_pair
    cores = os.cpu_count() or 1
    cfg = get_cfg()
    _set_infer_layers(cfg, infer_layers)
    cfg.DPN.MAX_DISP = max_disp
    w = hash_state_dict(build_model(cfg)[0].state_dict())
    ocfg = O.OracleCfg(num_infer_layers=infer_layers, max_disp=max_disp)

    def timed(l, r, nthr, reps):
        torch.set_num_threads(nthr)
        ts = []
        with torch.no_grad():
            for i in range(reps + 1):                               # first = warm-up
                t0 = time.perf_counter()
                O.forward(w, ocfg, l[None], r[None])
                ts.append(time.perf_counter() - t0)
        ts = sorted(ts[1:])
        return ts[len(ts) // 2] if len(ts) % 2 else ts[0]           # median of 5 or 3 / the better of 2

    # the all-core run is attempted only if it is not a collapse: on a 256-thread host the op-by-op path with every core is three
    # orders of magnitude SLOWER than with 16 threads (thread wake-ups dominate the small ATen ops); a micro-probe of one small
    # convolution + softmax (milliseconds) decides, and its two timings are reported either way
    n16 = min(cores, 16)
    probe = None
    if cores > n16:
        # (a probe of the forward itself is no good: at 256 threads one 96x192 forward took 142 s against 0.06 s on 16 threads)
        xs = torch.randn(1, 64, 24, 48)
        ws = torch.randn(64, 64, 3, 3)

        def micro(nthr):
            torch.set_num_threads(nthr)
            with torch.no_grad():
                for _ in range(3):
                    torch.softmax(torch.nn.functional.conv2d(xs, ws, padding=1), 1)
                t0 = time.perf_counter()
                for _ in range(20):
                    torch.softmax(torch.nn.functional.conv2d(xs, ws, padding=1), 1)
                return (time.perf_counter() - t0) / 20
        t_a, t_b = micro(n16), micro(cores)
        probe = {"op": "conv3x3 64->64 @24x48 + softmax", "threads_%d_ms" % n16: round(t_a * 1e3, 3), "threads_%d_ms" % cores: round(t_b * 1e3, 3)}
    all_cores_ok = probe is not None and probe["threads_%d_ms" % cores] <= 1.5 * probe["threads_%d_ms" % n16]

    def one_size(h, wd, with_1thread):
        l, r, _ = synthetic_pair(h, wd, seed=1000)
        rec = {"threads_%d_s" % n16: round(timed(l, r, n16, 5 if with_1thread else 3), 3)}      # the bench's own size: SURVEY 8(d)'s median of 5
        if cores > n16 and all_cores_ok:
            rec["threads_%d_s" % cores] = round(timed(l, r, cores, 2), 3)
        if with_1thread:
            torch.set_num_threads(1)
            with torch.no_grad():
                t0 = time.perf_counter()
                O.forward(w, ocfg, l[None], r[None])
                rec["threads_1_s"] = round(time.perf_counter() - t0, 3)
        return rec

    prev = torch.get_num_threads()
    try:
        main_rec = one_size(height, width, True)
        sizes = {"%dx%d" % (width, height): main_rec}
        for (h2, w2) in ((375, 1242), (540, 960)):
            if (h2, w2) != (height, width):
                sizes["%dx%d" % (w2, h2)] = one_size(h2, w2, False)
    finally:
        torch.set_num_threads(prev)
    best_key = min((k for k in main_rec if k != "threads_1_s"), key=lambda k: main_rec[k])
    nthr, dt = int(best_key.split("_")[1]), main_rec[best_key]
    return {"value": round(1.0 / dt, 4), "unit": "stereo pairs/s", "cores": nthr, "kind": "port",
            "value_1thread": round(1.0 / main_rec["threads_1_s"], 4), "host_cores": cores,
            "all_cores_probe": probe if probe is None else dict(probe, full_size_run=all_cores_ok),
            "seconds_per_forward": sizes,
            "pairs_per_s": {sz: {k[:-2]: round(1.0 / v, 4) for k, v in rec.items()} for sz, rec in sizes.items()},
            "sample": "one synthetic pair per size, batch 1 (oracle/nmrf_oracle.py, torch CPU fp32): 1 warm-up + median of 5 forwards on 16 "
                      "threads at the bench's own size (median of 3 at the other size) and -- unless a one-op probe shows the all-core run collapsing (all_cores_probe) -- 1 warm-up + "
                      "the better of 2 on all %d host cores; one forward on 1 thread at %dx%d; `value` = the faster thread count at "
                      "%dx%d (%d threads, %.2f s)" % (cores, width, height, width, height, nthr, dt)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(local_rank, world, port, argv):
    """One rank of a self-launched job (the counterpart of main.py:147-180 `_distributed_worker`)."""
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    run(parse(argv))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: no torchrun around us -> start the N workers ourselves (spawn, one per GPU); rank 0 prints
        if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus:
            raise SystemExit("--gpus %d but only %d visible" % (args.gpus, torch.cuda.device_count()))
        import torch.multiprocessing as mp
        mp.start_processes(_worker, args=(args.gpus, _free_port(), argv), nprocs=args.gpus, join=True, start_method="spawn")
        return
    run(args)


def run_plumbing(args, rank, world):
    """No GPU on this host: the launcher, the batch split and the result gather on `world` gloo ranks with a stand-in forward
    (the product has no CPU path and none is faked: the JSON line carries "valid": false)."""
    import torch.distributed as dist
    from nmrf_amd.parallel import OverlappedGather, shard_range
    from nmrf_amd.utils.hashinit import synthetic_pair
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    b = args.batch
    lo, hi = shard_range(world * b, rank, world)
    h, w = min(args.height, 48), min(args.width, 96)
    pairs = [synthetic_pair(h, w, seed=1000 + i)[:2] for i in range(lo, hi)]
    img1, img2 = torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])
    gath = OverlappedGather()
    step = lambda: gath.submit((img1.mean(1) - img2.mean(1)).abs())
    for _ in range(max(args.warmup, 1)):
        out = step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    want = torch.stack([(lambda p: (p[0].mean(0) - p[1].mean(0)).abs())(synthetic_pair(h, w, seed=1000 + i)[:2]) for i in range(world * b)])
    ok = bool(torch.equal(out, want))                       # every rank holds all disparities, in rank order
    # the config-4 leg of the GPU path (8 pairs per rank, contiguous split of the 8 * world job, same gather): plumbing only
    lo4, hi4 = shard_range(8 * world, rank, world)
    p4 = [synthetic_pair(h, w, seed=1000 + i)[:2] for i in range(lo4, hi4)]
    a4, b4 = torch.stack([p[0] for p in p4]), torch.stack([p[1] for p in p4])
    out4 = gath.submit((a4.mean(1) - b4.mean(1)).abs())
    want4 = torch.stack([(lambda p: (p[0].mean(0) - p[1].mean(0)).abs())(synthetic_pair(h, w, seed=1000 + i)[:2]) for i in range(8 * world)])
    ok4 = bool(torch.equal(out4, want4))
    if rank == 0:
        print(json.dumps({"metric": "stereo pairs/sec at %dx%d" % (args.width, args.height), "value": None,
                          "unit": "stereo pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "valid": False,
                          "config": {"workload": "PLUMBING ONLY: no GPU on this host, stand-in forward on %d gloo ranks (launcher, batch "
                                                 "split, result gather); not a measurement" % world,
                                     "global_batch": world * b, "parallelism": "batch-shard x%d" % world, "launch": "cpu stand-in",
                                     "result_gather": world > 1, "gather_correct": ok},
                          "config4": {"workload": "PLUMBING ONLY: 8 pairs per rank x %d" % world, "value": None, "global_batch": 8 * world,
                                      "gather_correct": ok4}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not (ok and ok4):
        raise SystemExit("gathered result differs from the single-process result")


def run(args):
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        if world == 1:
            raise SystemExit("bench.py needs an MI355X: the NMRF hot path has no CPU path (only `--gpus N`, N > 1, runs its "
                             "launcher / gather plumbing on gloo without one)")
        return run_plumbing(args, rank, world)
    import torch.distributed as dist
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    numa = None
    if world > 1:                                            # one process per GPU: host threads + pinned rings on the GPU's NUMA node
        from nmrf_amd.parallel import pin_to_gpu_numa       # (N = 1 keeps every core: the cpu_baseline leg wants them)
        numa = pin_to_gpu_numa(local_rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (only a rank started without a launcher gets here without MASTER_PORT: `--force-dist` at N = 1 -- a FREE port, not a fixed one:
        # a store of a previous run on the same port in TIME_WAIT made one such run of a gpu_round.sh fail to rendezvous)
        os.environ.setdefault("MASTER_PORT", str(_free_port()) if world == 1 else "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from nmrf_amd import kernels as K
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.parallel import OverlappedGather, gather_disparity
    from nmrf_amd.utils.hashinit import apply_hash_weights, synthetic_pair

    cfg = get_cfg()
    _set_infer_layers(cfg, args.infer_layers)
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

    overlapped = OverlappedGather(single_rank_too=args.force_dist)      # the gather rides a side stream behind the next step's compute

    def gather(disp):
        if use_dist and not args.no_gather:
            if world == 1 and args.sync_gather:               # --force-dist smoke: the collective on a 1-rank group
                g = torch.empty_like(disp)
                dist.all_gather_into_tensor(g, disp.contiguous())
                return g
            return gather_disparity(disp) if args.sync_gather else overlapped.submit(disp)
        return disp

    def timed_region(smp, steps, warmup):
        """`warmup` untimed steps (+ the hipGraph capture), then EXACTLY `steps` steps between barrier + synchronize on both sides;
        the MAX over ranks.  Returns (elapsed s, graph or None, the graph's static output, forward, step)."""
        def forward():
            return model(smp)["disp"]

        def step():
            return gather(forward())

        graph, static_out = None, None
        for _ in range(max(warmup, 1)):
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
            def run_step():
                graph.replay()
                return gather(static_out)
        else:
            run_step = step
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        overlapped.finish()                                   # every gather of the timed steps has landed inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, graph, static_out, forward, step

    def gather_alone_ms(d0):
        """the collective alone ("report the gather time separately"): 10 back-to-back all-gathers of one step's output"""
        alone = (lambda t: overlapped.submit(t)) if world == 1 else gather_disparity      # (1-rank group: same RCCL call)
        alone(d0)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for _ in range(10):
            alone(d0)
        overlapped.finish()
        g1.record()
        torch.cuda.synchronize()
        return g0.elapsed_time(g1) / 10

    with torch.no_grad():
        elapsed, graph, static_out, forward, step = timed_region(sample, args.steps, args.warmup)
        gather_ms = None
        if use_dist and not args.no_gather:
            gather_ms = gather_alone_ms(static_out if graph is not None else forward())

        # BASELINE config 4 (KITTI 1242x375, batch 64 over 8 GPUs = 8 pairs per GPU, nmrf/utils/evaluation.py:61-69's contiguous
        # split): measured NEXT TO the default weak-scaling line whenever the job has more than one rank (or --config4), never in
        # its place -- `value` stays the default line's.  Same timed_region, same gather, its own hipGraph at batch 8.
        config4 = None
        if (world > 1 or args.config4) and b != 8 and args.backbone == "resnet":
            try:
                from nmrf_amd.parallel import shard_range
                lo, hi = shard_range(8 * world, rank, world)
                prs = [synthetic_pair(args.height, args.width, seed=1000 + i)[:2] for i in range(lo, hi)]
                smp8 = {"img1": torch.stack([p[0] for p in prs]).to(dev), "img2": torch.stack([p[1] for p in prs]).to(dev)}
                k4 = max(3, args.steps // 4)
                dt4, g4, so4, fwd4, _ = timed_region(smp8, k4, 2)
                gms4 = gather_alone_ms(so4 if g4 is not None else fwd4()) if (use_dist and not args.no_gather) else None
                config4 = {"workload": "KITTI %dx%d, global batch %d = 8 pairs per GPU x %d (BASELINE config 4%s)" % (
                               args.width, args.height, 8 * world, world, "" if world == 8 else ": its per-GPU shard at this N"),
                           "value": round(8 * world * k4 / dt4, 3), "unit": "stereo pairs/s", "steps": k4, "warmup": 2,
                           "ms_per_step": round(dt4 / k4 * 1e3, 3), "global_batch": 8 * world,
                           "launch": "hipGraph" if g4 is not None else "eager",
                           "gather_ms_alone": None if gms4 is None else round(gms4, 4),
                           "gather_bytes_per_rank_per_step": 8 * args.height * args.width * 4}
                del smp8, g4, so4
            except Exception as e:                              # never at the expense of the default line
                config4 = {"error": repr(e)}

        # dominant hand-written kernel, timed live with HIP events on its launching stream (eager launches,
        # same inputs, right after the timed region so clocks/caches are in the same state)
        # The side stream that overlaps the conv heads with the proposal stage is switched off for this pass (and for
        # the rocprofv3 run of tools/gpu_round.sh): a kernel sharing the chip with a MIOpen conv reports the conv's
        # duration, not its own (stripe launches of 0.42 ms measured as 4.7 ms at batch 8).
        prev_overlap = os.environ.get("NMRF_OVERLAP")
        os.environ["NMRF_OVERLAP"] = "0"
        timer.enabled = True
        n_timed_fwd = max(3, min(args.steps, 10))
        # (the arguments of one launch of the pair kernel are kept: the clock sampling below replays exactly that launch)
        pair_call, pair_orig = [], K.nmp_block_pair

        def _pair_tap(*a, **kw):
            if not pair_call and len(a) > 1 and a[1] is not None:
                pair_call.append((a, kw))
            return pair_orig(*a, **kw)
        K.nmp_block_pair = _pair_tap
        try:
            for _ in range(n_timed_fwd):
                step()
        finally:
            K.nmp_block_pair = pair_orig
        overlapped.finish()
        torch.cuda.synchronize()
        timer.enabled = False
        if prev_overlap is None:
            del os.environ["NMRF_OVERLAP"]
        else:
            os.environ["NMRF_OVERLAP"] = prev_overlap
        kstats = timer.stats()

        # The shader clock under the dominant kernel: the peaks of MI355X_MICROARCH.md are quoted at the 2.4 GHz boost
        # clock, under matrix-heavy kernels the chip holds less.  Every block of the pair kernel records its CU's shader-clock counter and
        # the chip's 100 MHz counter at entry and exit (nmrf_nmp_block16_clock_records) during 320 back-to-back launches of the launch
        # captured above; the records of the last launch are read.
        clocks = None
        if pair_call and not args.no_clock_sample:
            try:
                pa, pkw = pair_call[0]
                kh, K.kernel_hook = K.kernel_hook, None
                try:
                    pair_orig(*pa, **pkw)
                    with K.BlockKernelClock(dev) as cs:
                        for _ in range(320):
                            pair_orig(*pa, **pkw)
                finally:
                    K.kernel_hook = kh
                clocks = {"nmp_block_pair": round(cs.ghz, 4), "min_max_over_blocks": [round(cs.ghz_min, 4), round(cs.ghz_max, 4)],
                          "blocks": cs.blocks, "block_life_us": round(cs.block_us, 2),
                          "how": "s_memtime per s_memrealtime (100 MHz) between entry and exit of every block of the last of 320 "
                                 "back-to-back launches of the dominant kernel (nmrf_nmp_block16_clock_records), summed over its blocks"}
            except Exception as e:
                clocks = {"error": repr(e)}

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
        stream_rec, stream_b8 = None, None
        if world == 1 and not args.no_stream_figure:
            from nmrf_amd.driver import StereoStream

            def stream_figure(bs, dtype, inflight=None):
                n_pairs = max(8 * bs, min(64, 4 * args.steps))
                host_pairs = [(i,) + tuple(t.cpu().to(dtype) for t in pairs[i % len(pairs)]) for i in range(n_pairs)]
                drv = StereoStream(model, dev, batch=bs, graph=not args.no_graph, inflight=inflight)
                list(drv.run(iter(host_pairs[:2 * bs])))                       # warm-up: buffers, hipGraph capture
                rates = []
                for _ in range(4):                                              # the FIRST full run of a fresh stream object is slower by up
                    torch.cuda.synchronize()                                    # to 27 % (host-side first touches of the pinned rings): it is
                    t1 = time.perf_counter()                                    # reported on its own, `value` = the median of the next three
                    n_done = sum(1 for _ in drv.run(iter(host_pairs)))
                    torch.cuda.synchronize()
                    rates.append(n_done / (time.perf_counter() - t1))
                return {"value": round(sorted(rates[1:])[1], 2), "first_run": round(rates[0], 1), "runs": [round(r, 1) for r in rates[1:]],
                        "unit": "stereo pairs/s", "pairs": n_done, "batch": bs,
                        "host_dtype": str(dtype).replace("torch.", ""), "launch": "hipGraph" if drv.use_graph else "eager",
                        "forwards_in_flight": drv.inflight}
            try:
                K.kernel_hook = None
                stream_rec = stream_figure(b, torch.uint8)
                stream_rec["note"] = ("nmrf_amd.driver.StereoStream: fresh HOST inputs per batch (uint8 images, as decoded from disk), persistent "
                                      "pinned rings, H2D / D2H on their own streams overlapped with compute, one hipGraph replay per batch, "
                                      "results back in host memory; `value` above is compute-only (hipGraph replay on resident inputs)")
                stream_rec["float32_host_images"] = stream_figure(b, torch.float32)["value"]
                if b == 1 and not args.no_graph:
                    stream_rec["two_forwards_in_flight"] = stream_figure(b, torch.uint8, inflight=2)["value"]
                    stream_rec["note"] += ("; `two_forwards_in_flight`: batches dealt to two replicas of the model, each with its own hipGraph and "
                                           "compute stream (StereoStream(inflight=2), off by default)")
                if b == 1 and (args.height, args.width) == (375, 1242) and args.backbone == "resnet":
                    stream_b8 = stream_figure(8, torch.uint8)                  # the batch the driver defaults to (N1)
            except Exception as e:
                stream_rec = {"error": repr(e)}
            K.kernel_hook = timer

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
        flops, nbytes = timer.mean_meta(k, "flops"), timer.mean_meta(k, "bytes")
        mfma_peak = round(SPLIT_MFMA_PEAK_TFLOPS, 1) if meta.get("split") else FP32_MFMA_PEAK_TFLOPS
        # a kernel is priced on its TIGHTER bound: the roofline (matrix pipe or HBM) its algorithmic work sits closer to
        f_mfma = flops / (ms * 1e-3) / 1e12 / mfma_peak if flops else None
        f_hbm = nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if nbytes else None
        bound = meta.get("bound", "mfma")
        if f_mfma is not None and f_hbm is not None:
            bound = "mfma" if f_mfma >= f_hbm else "hbm"
        if bound == "mfma":
            ach, peak, unit = flops / (ms * 1e-3) / 1e12, mfma_peak, "TFLOP/s"
        else:
            ach, peak, unit = nbytes / (ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        pr = pmc_rec(meta.get("pmc")) if pmc_ok else {}
        rec = {"bound": bound, "kernel": meta.get("label", k), "row": meta.get("row"), "achieved": round(ach, 3), "peak": peak,
               "unit": unit, "frac": round(ach / peak, 4), "traffic": pr.get("hbm_bytes"),
               "launch_ms": round(ms, 4), "launches_timed": cnt, "ms_per_forward": round(ms * cnt / n_timed_fwd, 4),
               "flop_per_launch": flops, "bytes_per_launch": nbytes}
        if f_mfma is not None and f_hbm is not None:
            rec["frac_other_bound"] = {"bound": "hbm" if bound == "mfma" else "mfma", "frac": round(f_hbm if bound == "mfma" else f_mfma, 4)}
        if bound == "mfma":
            rec["pipe"] = ("fp16 MFMA, split fp32 operands: peak = 2500 TFLOP/s / 3 products (csrc/split_mfma.h)" if meta.get("split")
                           else "fp32 MFMA")
        if pr:
            rec["traffic_source"] = "profiles/pmc_traffic.json: " + str(pmc.get("_source", "?"))
            for key in ("mfma_busy", "lds_bank_conflict_ratio"):
                if key in pr:
                    rec[key] = pr[key]
        if isinstance(clocks, dict) and clocks.get("nmp_block_pair") and k.startswith("nmp_block") and bound == "mfma":
            # the matrix peak scales with the clock (2.4 GHz is what MI355X_MICROARCH.md's figure assumes); measured inside the pair
            # kernel, applied to the block kernels (the same instruction mix; profiles/r04k_block_timeline.txt saw the same clock)
            rec["sustained_clock_ghz"] = clocks["nmp_block_pair"]
            rec["frac_at_sustained_clock"] = round(ach / (peak * clocks["nmp_block_pair"] / BOOST_CLOCK_GHZ), 4)
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
                       "result_gather": bool(use_dist and not args.no_gather),
                       "numa_pinning_rank0": numa,
                       "gather": None if gather_ms is None else {
                           "ms_alone": round(gather_ms, 4), "placement": "on the compute stream" if args.sync_gather else
                           "side stream behind the next step (nmrf_amd.parallel.OverlappedGather); all gathers complete inside the timed region"}},
            "hot_path_ms": None if hp_ms is None else round(hp_ms, 3),
            "config4": config4,
            "roofline": roof,
            "other_kernels": others,
            "sustained_clock_ghz": clocks,
        }
        if stream_rec is not None:
            res["stream_end_to_end"] = stream_rec
        if stream_b8 is not None:
            res["stream_end_to_end_b8"] = stream_b8
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
