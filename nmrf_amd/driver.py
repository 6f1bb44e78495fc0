"""Batched inference driver: the counterpart of the reference's demo / submission loop
(inference.py:61-126) and of the timing convention of nmrf/utils/evaluation.py:203-267, without its batch-1
synchronous structure: pairs are grouped into batches, staged in pinned host memory, copied H2D on a side stream
while the previous batch computes, and the finished disparities come back D2H asynchronously.

    python -m nmrf_amd.driver --left 'L/*.png' --right 'R/*.png' --output out/ [--ckpt kitti.pth] [--batch 8]

Disparity maps are written as KITTI 16-bit PNGs (uint16 = round(disp * 256), nmrf/utils/frame_utils.py:237-239)
or .npy.  Works with any callable `model(sample) -> {'disp': [B,H,W]}`; `model=None` builds nmrf_amd's NMRF.
"""
import argparse
import glob
import os
import sys
import time

import numpy as np
import torch

from . import kernels as K


def encode_kitti_disp(disp):
    """float disparity [H,W] -> uint16 array, KITTI convention (0 = invalid is never produced here)."""
    return np.clip(np.round(np.asarray(disp, dtype=np.float64) * 256.0), 0, 65535).astype(np.uint16)


def load_rgb(path):
    """-> uint8 [3,H,W] in 0..255 (the reference feeds the same values as float, nmrf/data/datasets.py:54-62; the conversion
    happens on the GPU, after the bytes have crossed PCIe)."""
    from PIL import Image
    img = np.array(Image.open(path).convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(img).permute(2, 0, 1).contiguous()


def batches(items, size):
    """Group consecutive items of equal image size AND dtype (of both views) into batches of at most `size` (order preserved):
    a batch shares one pinned staging buffer, so a uint8 pair never lands in a float32 group."""
    cur, shape = [], None
    for it in items:
        s = (tuple(it[1].shape), it[1].dtype, tuple(it[2].shape), it[2].dtype)
        if s[0] != s[2] or s[1] != s[3]:
            raise ValueError("left and right view of %r differ in shape or dtype: %s %s vs %s %s" % (it[0], s[0], s[1], s[2], s[3]))
        if cur and (s != shape or len(cur) == size):
            yield cur
            cur = []
        cur.append(it)
        shape = s
    if cur:
        yield cur


def _replica(model):
    """A private copy of the model for one more lane: own parameters, own lazily-built launch state (packed weights, persistent
    activation grids, side stream).  Stream objects are not copied (the replica creates its own on first use)."""
    import copy
    memo = {}
    for m in (model.modules() if hasattr(model, "modules") else ()):
        for v in vars(m).values():
            if isinstance(v, (torch.cuda.Stream, torch.cuda.Event)):
                memo[id(v)] = None
    with torch.no_grad():
        rep = copy.deepcopy(model, memo)
    if hasattr(rep, "eval"):
        rep.eval()
    if hasattr(rep, "range_check"):
        rep.range_check = False           # the stream reads the (device-wide) range flag once per drained batch
    return rep


class _Plan:
    """Everything StereoStream keeps per (image shape, dtype): a ring of pinned host buffers and device staging buffers for the
    inputs, the static input / output tensors of ONE captured hipGraph of model.forward, and a ring of device + pinned host
    buffers for the results."""

    def __init__(self, owner, shape, dtype):
        dev, b, depth = owner.device, owner.batch, owner.depth
        c, h, w = shape
        self.shape, self.dtype = shape, dtype
        self.pin_in = [torch.empty(2, b, c, h, w, dtype=dtype).pin_memory() for _ in range(depth)]
        self.dev_in = [torch.empty(2, b, c, h, w, dtype=dtype, device=dev) for _ in range(depth)]
        self.static_in = torch.empty(2, b, c, h, w, dtype=dtype, device=dev)
        self.graph, self.static_out = None, None
        self.out_dev, self.pin_out = [None] * depth, [None] * depth
        self.pin_flag = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.ev_in = [torch.cuda.Event() for _ in range(depth)]        # H2D of the slot finished
        self.ev_used = [None] * depth                                   # compute has consumed dev_in[slot]
        self.ev_out = [torch.cuda.Event() for _ in range(depth)]       # out_dev[slot] written
        self.ev_d2h = [None] * depth                                    # pin_out[slot] complete

    def sample(self, n=None):
        x = self.static_in if n is None else self.static_in[:, :n]
        return {"img1": x[0], "img2": x[1]}


class StereoStream:
    """Pipelined batches: H2D of batch i+1 and D2H of batch i-1 run on their own streams while batch i computes.

    * images travel as they are handed over: uint8 (decoded PNGs: 1.4 MB per KITTI view over PCIe, converted inside the staging
      kernel nmrf_prep_images_s2d_u8) or float32 (4x the bytes);
    * host staging buffers are pinned ONCE per image shape and reused (a ring of `depth` slots), results come back into a ring of
      pinned buffers -- no per-batch torch.stack / pin_memory;
    * the forward is ONE hipGraph per (shape, batch), captured on first use and replayed on static device buffers (a short final
      batch is padded with copies of its last pair and the padding discarded): per batch the host issues two copies and a replay
      instead of ~150 kernel launches.  graph=False, a model that cannot be captured, or a CPU device -> eager calls.
    Results are yielded in input order as CPU tensors.  `copy_out=False` hands out VIEWS into the pinned result ring (and runs on
    the caller's thread): batch i's slot is re-armed when batch i + depth is enqueued, which happens just before the first item of
    batch i + depth - 1 is yielded -- so a view of batch i is valid until the consumer asks for the first item of batch
    i + depth - 1 (the ring is at least three deep in this mode: a view survives the whole next batch).  Every line the consumer
    reads of a view stays in the CPU cache and slows the next D2H into that slot down, see nmrf_host_read_evict; the default
    hands out pageable copies and evicts.
    Host-side staging uses non-temporal stores (nmrf_host_copy_nt): a DMA that has to snoop freshly written lines out of a CPU
    cache costs ~20 ms per batch on this platform, whatever the batch size (tools/driver_probe3.py)."""

    def __init__(self, model, device="cuda", batch=8, graph=True, depth=2, copy_out=True, threaded=True, inflight=None):
        """inflight: forwards that may run on the GPU AT THE SAME TIME (default 1).  With inflight = 2 the stream deals batches to two
        LANES -- each a replica of the model (weights copied once: 24 MB) with its own activations, its own captured hipGraph and its
        own compute stream -- so that the encoder of pair i + 1 runs beside the message-passing stages of pair i (inference.py:61-100 /
        evaluation.py:203-267 process one pair at a time); results are still yielded in input order and are bit-equal to one lane's.
        Measured at KITTI batch 1 on the MI355X (profiles/r04f_driver_lanes.txt): 249.8 pairs/s with two lanes against 285.6 with one
        (compute-only 309.3 on that box) -- the two forwards' one-workgroup-per-CU kernels (156 KB of LDS each) cannot share a CU and
        everything that does overlap runs slower than in turn -- so one lane stays the default; the option is kept for small images."""
        self.model, self.device, self.batch = model, torch.device(device), batch
        self.on_gpu = self.device.type == "cuda"
        self.depth, self.copy_out, self.use_graph = max(2 if copy_out else 3, depth), copy_out, graph and self.on_gpu
        self._inflight_req = inflight
        # threaded: staging + launches of batch i+1 / i+2 on a producer thread while the caller's thread drains batch i (the staging
        # copies and the evicting read-out run outside the GIL).  Needs a third slot (see _run_threaded) and handed-out COPIES.
        self.threaded = bool(threaded) and self.on_gpu and copy_out
        inflight = self._inflight_req
        if inflight is None:
            inflight = 1
        self.inflight = max(1, int(inflight)) if self.threaded else 1    # (lanes are fed by the producer thread)
        if self.threaded:
            self.depth = max(3, self.depth)
            self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.plans = {}
        self._lane_models, self._lane_streams = [model], [None]          # lane 0: the caller's model on the current stream
        # models of this package defer their fp16 range check to the driver (one flag read per drained batch) -- only while run()
        # is active: the model's own check is switched back on when the stream ends (run()'s finally)
        self._range_model = model if (self.on_gpu and hasattr(model, "range_check")) else None
        if self.on_gpu:
            self.h2d = torch.cuda.Stream(self.device)
            self.d2h = torch.cuda.Stream(self.device)
            for _ in range(1, self.inflight):
                self._lane_models.append(_replica(model))
                self._lane_streams.append(torch.cuda.Stream(self.device))

    # ---- GPU path ------------------------------------------------------------------------------------------------------
    def _plan(self, group, lane=0):
        t = group[0][1]
        key = (lane, tuple(t.shape), t.dtype)
        plan = self.plans.get(key)
        if plan is None:
            plan = self.plans[key] = _Plan(self, tuple(t.shape), t.dtype)
            plan.lane = lane
        return plan

    def _capture(self, plan):
        """Two eager forwards (weight packing, function attributes, allocator warm-up), then the capture."""
        sample = plan.sample()
        model = self._lane_models[plan.lane]
        try:
            with torch.no_grad():
                for _ in range(2):
                    model(sample)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = model(sample)["disp"]
            plan.graph, plan.static_out = g, out
        except Exception as e:                                  # not capturable (e.g. a model with host syncs): eager from now on
            print("[nmrf_amd.driver] hipGraph capture failed (%s); eager launches" % str(e).splitlines()[0], file=sys.stderr)
            torch.cuda.synchronize(self.device)
            self.use_graph = False

    def _enqueue(self, plan, group, slot):
        lane_stream = self._lane_streams[plan.lane]
        if lane_stream is None:
            return self._enqueue_on_current(plan, group, slot)
        with torch.cuda.stream(lane_stream):
            return self._enqueue_on_current(plan, group, slot)

    def _enqueue_on_current(self, plan, group, slot):
        n = len(group)
        pin = plan.pin_in[slot]
        plan.ev_in[slot].synchronize()                          # the slot's previous H2D has left the pinned buffer (long ago)
        for j, (_, left, right) in enumerate(group):
            K.host_copy_nt(pin[0, j], left)
            K.host_copy_nt(pin[1, j], right)
        for j in range(n, self.batch if self.use_graph else n): # pad a short batch for the fixed-size graph
            K.host_copy_nt(pin[0, j], group[-1][1])
            K.host_copy_nt(pin[1, j], group[-1][2])
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.h2d):
            if plan.ev_used[slot] is not None:
                self.h2d.wait_event(plan.ev_used[slot])         # dev_in[slot] has been consumed by its previous batch
            plan.dev_in[slot].copy_(pin, non_blocking=True)
            plan.ev_in[slot].record(self.h2d)
        main.wait_event(plan.ev_in[slot])
        if self.use_graph and plan.graph is None:
            plan.static_in.copy_(plan.dev_in[slot])
            self._capture(plan)
        with torch.no_grad():
            if self.use_graph and plan.graph is not None:
                plan.static_in.copy_(plan.dev_in[slot], non_blocking=True)
                used = torch.cuda.Event()
                used.record(main)
                plan.graph.replay()
                disp = plan.static_out
            else:
                x = plan.dev_in[slot]
                disp = self._lane_models[plan.lane]({"img1": x[0, :n], "img2": x[1, :n]})["disp"]
                used = torch.cuda.Event()
                used.record(main)
        plan.ev_used[slot] = used
        if plan.out_dev[slot] is None or plan.out_dev[slot].shape[1:] != disp.shape[1:]:
            plan.out_dev[slot] = torch.empty((self.batch,) + tuple(disp.shape[1:]), dtype=disp.dtype, device=self.device)
            plan.pin_out[slot] = torch.empty((self.batch,) + tuple(disp.shape[1:]), dtype=disp.dtype).pin_memory()
        if plan.ev_d2h[slot] is not None:
            main.wait_event(plan.ev_d2h[slot])                  # the slot's previous result has left out_dev[slot]
        plan.out_dev[slot][:disp.shape[0]].copy_(disp, non_blocking=True)
        plan.ev_out[slot].record(main)
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(plan.ev_out[slot])
            plan.pin_out[slot].copy_(plan.out_dev[slot], non_blocking=True)
            if self._range_model is not None:               # the sticky fp16 range flag rides back with the results (4 bytes)
                plan.pin_flag[slot].copy_(K.range_flag(self.device), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.d2h)
        plan.ev_d2h[slot] = done
        return [g[0] for g in group], plan.pin_out[slot], done, plan.pin_flag[slot]

    def _drain(self, pending):
        keys, host, done = pending[:3]
        if done is not None:
            done.synchronize()
            if len(pending) > 3 and int(pending[3]) != 0:   # the fp16 range flag as it stood when this batch had finished
                K.check_range(self.device)                  # raises NmrfHipError (and clears the flag)
        for k, d in zip(keys, host):
            yield k, (K.host_read_evict(d) if self.copy_out and done is not None else d)

    def _run_threaded(self, pairs):
        """Producer thread: decode (the `pairs` iterator), stage, enqueue.  Caller's thread: wait, read out, yield.  One-element queue:
        while the caller drains batch j the producer may have enqueued j+1 (queued) and be working on j+2 -- never on j+3, which
        shares batch j's slot in the three-slot ring, before batch j+1 has been taken, i.e. before batch j is fully read out."""
        import queue
        import threading
        # `inflight` batches may be queued behind the one being drained; the rings are deep enough for it (see _slot)
        q, stop = queue.Queue(maxsize=self.inflight), threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def produce():
            try:
                torch.cuda.set_device(self._dev_index)
                for i, group in enumerate(batches(pairs, self.batch)):
                    lane, slot = self._slot(i)
                    if stop.is_set() or not put(self._enqueue(self._plan(group, lane), group, slot)):
                        return
                put(None)
            except BaseException as e:                          # surfaces in the caller's thread
                put(e)

        th = threading.Thread(target=produce, name="nmrf-stage", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield from self._drain(item)
        finally:
            stop.set()
            th.join()
            torch.cuda.synchronize(self.device)                 # nothing of an abandoned batch stays in flight over the rings

    def run(self, pairs):
        """pairs: iterable of (key, left [3,H,W], right [3,H,W]) (uint8 or float32, 0..255) -> yields (key, disparity [H,W] CPU
        tensor) in input order."""
        # The per-forward range check (a host sync) is deferred to the stream's own check while a run is active.  Runs are COUNTED on
        # the model (two streams interleaved on one model, or a generator that is dropped half-way and collected later, must not
        # restore the wrong value): the caller's setting is saved by the first active run and restored by the last one to end.
        m = self._range_model
        if m is not None:
            if getattr(m, "_stream_runs", 0) == 0:
                m._range_check_saved = m.range_check
            m._stream_runs = getattr(m, "_stream_runs", 0) + 1
            m.range_check = False
        try:
            yield from (self._run_threaded(pairs) if self.threaded else self._run_inline(pairs))
        finally:
            if m is not None:
                m._stream_runs -= 1
                if m._stream_runs == 0:
                    m.range_check = m._range_check_saved

    def _slot(self, i):
        """batch i -> (lane, ring slot of that lane's plan).  The producer is at most `inflight` + 1 batches ahead of the batch being
        drained, and batch i shares its (lane, slot) with batch i - inflight * depth: depth >= 3 keeps them apart."""
        return i % self.inflight, (i // self.inflight) % self.depth

    def _run_inline(self, pairs):
        pending, i = None, 0
        for group in batches(pairs, self.batch):
            if self.on_gpu:
                cur = self._enqueue(self._plan(group, 0), group, i % self.depth)
            else:
                with torch.no_grad():
                    disp = self.model({"img1": torch.stack([g[1] for g in group]), "img2": torch.stack([g[2] for g in group])})["disp"]
                cur = ([g[0] for g in group], disp, None)
            i += 1
            if pending is not None:                              # batch i is queued: now wait for batch i-1 and hand it out
                yield from self._drain(pending)
            pending = cur
        if pending is not None:
            yield from self._drain(pending)


def build_default_model(ckpt=None, opts=(), device="cuda"):
    from .config import get_cfg
    from .models import build_model
    cfg = get_cfg()
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    model = build_model(cfg)[0].eval()
    if ckpt:
        sd = torch.load(ckpt, map_location="cpu")
        model.load_state_dict(sd.get("model", sd), strict=cfg.SOLVER.STRICT_RESUME)
    return model.to(device)


def main(argv=None):
    ap = argparse.ArgumentParser(description="NMRF-Stereo batched inference on MI355X")
    ap.add_argument("--left", required=True, help="glob of left images")
    ap.add_argument("--right", required=True, help="glob of right images")
    ap.add_argument("--output", required=True)
    ap.add_argument("--ckpt", default=None, help="reference checkpoint (.pth, {'model': state_dict} or bare)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--format", choices=("kitti16", "npy"), default="kitti16")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE config overrides")
    args = ap.parse_args(argv)
    lefts, rights = sorted(glob.glob(args.left)), sorted(glob.glob(args.right))
    assert lefts and len(lefts) == len(rights), "need as many left as right images"
    os.makedirs(args.output, exist_ok=True)
    model = build_default_model(args.ckpt, args.opts)
    pairs = ((os.path.splitext(os.path.basename(l))[0], load_rgb(l), load_rgb(r)) for l, r in zip(lefts, rights))
    t0, n = time.perf_counter(), 0
    for key, disp in StereoStream(model, batch=args.batch).run(pairs):
        if args.format == "npy":
            np.save(os.path.join(args.output, key + ".npy"), disp.numpy())
        else:
            from PIL import Image
            Image.fromarray(encode_kitti_disp(disp.numpy())).save(os.path.join(args.output, key + ".png"))
        n += 1
    dt = time.perf_counter() - t0
    print("%d pairs in %.2f s (%.1f pairs/s incl. image I/O)" % (n, dt, n / dt), file=sys.stderr)


if __name__ == "__main__":
    main()
