"""N4: one optimisation step on the MI355X, shaped like the reference's training loop (main.py:403-430).  Two extents
(nmrf_amd.models.NMRF.enable_grad_slice):
  * the SLICE (default): the WHOLE inference and refinement stages with their three heads, and -- behind the proposal loss -- the WHOLE
    propagation stage with its head, and the seed filter: 315 of the model's 340 parameter tensors, the convolutional modules frozen on their fused
    forward-only kernels;
  * full=True: every parameter -- encoder, matching heads and DPN context convolutions on stock PyTorch-ROCm autograd, joined to the HIP
    stages by the backward of the cost volume, the cost taps and the warp + correlation rows.

    model.train(); loss_dict = criterion(model(sample), sample); losses = sum_k weight_dict[k] * loss_dict[k]
    param.grad = None; losses.backward(); clip_grad_norm_(GRAD_CLIP); optimizer.step()

and, with more than one rank, the gradient average DistributedDataParallel performs for the reference (main.py:334-339): bucket by bucket
under the backward pass (`OverlappedGradientReducer`: ~8 MB buckets in reverse parameter order, asynchronous all-reduces over RCCL,
identical collective order on every rank), or as ONE flat all-reduce behind it (`allreduce_gradients`: the first step, and callers
without a reducer); one process per GPU, weights replicated, batch sharded as in nmrf_amd.parallel."""
import torch
import torch.distributed as dist

# the WHOLE inference, refinement and propagation stages (attention backward kernels, round 5) and the four prediction heads
SLICE_PREFIXES = ("infer_head.", "infer_score_head.", "refine_head.", "inference.", "refinement.", "dpn.prop_head.", "dpn.propagation.",
                  "dpn.mlp.")                        # + the seed filter (Conv1d x 3 + softmax: the `init` loss)


def slice_parameters(model):
    """[(name, parameter)] the gradient graph reaches, in named_parameters() order: the slice, or with enable_grad_slice(full=True)
    every parameter."""
    if getattr(model, "grad_full", False):
        return list(model.named_parameters())
    return [(name, p) for name, p in model.named_parameters() if name.startswith(SLICE_PREFIXES)]


def build_slice_optimizer(model, cfg):
    """AdamW with the reference's parameter groups, in the reference's ORDER (build_optimizer, main.py:186-245; an optimizer state dict
    stores its groups by position, so `checkpoint_latest.pth` of either side loads into the other's optimizer) over the parameters the
    gradient graph reaches: plain parameters at BASE_LR / WEIGHT_DECAY; modules named `*sampling_offsets*` (the MSDeformAttn offsets of
    the Swin-T neck) at BASE_LR x 0.1 (main.py:217-218, 227-228); normalisation layers at WEIGHT_DECAY_NORM; a Swin trunk
    (`image_encoder.backbone.*`) at BASE_LR x BACKBONE_LR_DECAY / BACKBONE_WEIGHT_DECAY, its bias tables without decay; the
    relative-position tables of the window attention without weight decay.  Parameters outside the gradient graph are frozen
    (requires_grad False: no kernel could fill their .grad); a parameter the caller froze beforehand STAYS frozen and joins no group,
    as in the reference (main.py:206-207)."""
    keep = {id(p) for _, p in slice_parameters(model)}
    for p in model.parameters():
        if id(p) not in keep:
            p.requires_grad_(False)
    s = cfg.SOLVER
    norm_types = (torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d, torch.nn.LayerNorm)
    plain, offsets, norms, trunk, trunk_tab, enc_tab, seen = [], [], [], [], [], [], set()
    for mname, module in model.named_modules():
        for pname, p in module.named_parameters(recurse=False):
            if id(p) not in keep or not p.requires_grad or id(p) in seen:
                continue
            seen.add(id(p))
            if ("%s.%s" % (mname, pname)).startswith("image_encoder.backbone"):
                (trunk_tab if "relative_position_bias_table" in pname else trunk).append(p)
            elif "sampling_offsets" in mname:
                offsets.append(p)
            elif "relative_position_enc_table" in pname:
                enc_tab.append(p)
            elif isinstance(module, norm_types) and s.WEIGHT_DECAY_NORM is not None:
                norms.append(p)
            else:
                plain.append(p)
    groups = [{"params": plain, "lr": s.BASE_LR},
              {"params": offsets, "lr": s.BASE_LR * 0.1},
              {"params": norms, "lr": s.BASE_LR, "weight_decay": s.WEIGHT_DECAY_NORM},
              {"params": trunk, "lr": s.BASE_LR * s.BACKBONE_LR_DECAY, "weight_decay": s.BACKBONE_WEIGHT_DECAY},
              {"params": trunk_tab, "lr": s.BASE_LR * s.BACKBONE_LR_DECAY, "weight_decay": 0.0},
              {"params": enc_tab, "lr": s.BASE_LR, "weight_decay": 0.0}]
    return torch.optim.AdamW([g for g in groups if g["params"]], weight_decay=s.WEIGHT_DECAY)


build_optimizer = build_slice_optimizer          # the reference's name (main.py:186): build_optimizer(model, cfg)


def allreduce_gradients(params, group=None):
    """Average the gradients of `params` over the ranks of `group` with ONE all-reduce of a flat bucket (what DDP's reducer does
    bucket by bucket, main.py:334-339); identity without a process group.  A parameter whose .grad is None on EVERY rank keeps None
    (AdamW then skips it, as with one rank: e.g. the propagation stage under the reference's weight_dict, which drops loss_prop) --
    decided by a second, tiny all-reduce of a has-gradient mask, so that the result does not depend on the world size; a parameter
    some rank has a gradient for takes zeros from the others."""
    params = [p for p in params if p.requires_grad]
    if not params or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    ref = next((p.grad for p in params if p.grad is not None), params[0])
    has = torch.tensor([0.0 if p.grad is None else 1.0 for p in params], device=ref.device, dtype=torch.float32)
    dist.all_reduce(has, op=dist.ReduceOp.SUM, group=group)
    live = [p for p, h in zip(params, has.tolist()) if h > 0]
    if not live:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in live])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for p in live:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p).clone()
        off += n
    return flat.numel()


class OverlappedGradientReducer:
    """The gradient average of DistributedDataParallel's reducer (main.py:334-339) overlapped with the backward pass: the parameters
    that carry gradients are cut into buckets of ~`bucket_bytes` in REVERSE registration order (the order their gradients become
    ready), every parameter's post-accumulate hook counts its bucket down, and a complete bucket is all-reduced asynchronously (RCCL
    works on its own stream) while autograd keeps producing the next one; `finish()` launches what is left, waits, averages and
    writes the gradients back.

    Every rank issues the SAME collectives in the SAME order whatever its data did:
      * the first step is the flat path (`allreduce_gradients`), whose has-gradient mask -- summed over the ranks -- defines the live
        parameter set once, identically everywhere (a loss term without weight leaves whole stages without gradients);
      * buckets are launched strictly in bucket order (b only after b - 1), early when complete, otherwise in finish();
      * a live parameter that got no gradient on this rank in this step contributes zeros.
    A gradient on a parameter outside the live set (the loss structure changed) raises: rebuild the reducer.
    One rank / no process group: everything is the identity."""

    def __init__(self, params, group=None, bucket_bytes=8 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.group, self.bucket_bytes = group, int(bucket_bytes)
        self.active = bool(self.params) and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.buckets = None                  # list of lists of parameters, set by the first (flat) step
        self._hooks, self._inflight, self._next, self._pending, self._bucket_of = [], [], 0, [], {}

    def _build(self, live):
        self.buckets, cur, size = [], [], 0
        for p in reversed(live):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): b for b, ps in enumerate(self.buckets) for p in ps}
        live_ids = set(self._bucket_of)
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._ready if id(p) in live_ids else self._stray))

    def _stray(self, p):
        raise RuntimeError("a parameter outside the reducer's live set received a gradient: the loss structure changed -- build a "
                           "new OverlappedGradientReducer")

    def prepare(self):
        """Before backward: arm the bucket counters."""
        if self.active and self.buckets is not None:
            self._pending = [len(ps) for ps in self.buckets]
            self._inflight, self._next = [], 0

    def _launch_ready(self, force=False):
        while self._next < len(self.buckets) and (force or self._pending[self._next] == 0):
            ps = self.buckets[self._next]
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
            self._inflight.append((ps, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            self._next += 1

    def _ready(self, p):
        self._pending[self._bucket_of[id(p)]] -= 1
        self._launch_ready()

    def finish(self):
        """After backward: the averaged gradients are in .grad when this returns.  Returns the number of floats reduced."""
        if not self.active:
            return 0
        if self.buckets is None:                                         # first step: the flat path defines the live set
            n = allreduce_gradients(self.params, self.group)
            self._build([p for p in self.params if p.grad is not None])
            return n
        self._launch_ready(force=True)
        world, n = dist.get_world_size(self.group), 0
        for ps, flat, work in self._inflight:
            work.wait()
            flat /= world
            off = 0
            for p in ps:
                k = p.numel()
                p.grad = flat[off:off + k].view_as(p).clone()
                off += k
            n += off
        self._inflight = []
        return n

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def train_step(model, criterion, optimizer, sample, grad_clip=1.0, group=None, reducer=None):
    """One step of main.py:413-430 on the parameters of `optimizer` (build_slice_optimizer).  sample: {'img1', 'img2', 'disp', 'valid'} (H, W multiples of
    DATASETS.DIVIS_BY: the training-mode forward does not pad).  reducer: an OverlappedGradientReducer over the optimizer's parameters
    (fit() builds one when there is more than one rank) -- the gradient average then runs bucket by bucket under the backward pass
    instead of as one flat all-reduce behind it.  Returns (total loss as a float, the loss dict)."""
    if not getattr(model, "grad_slice", False):
        raise RuntimeError("call model.train().enable_grad_slice() first: without it the training-mode forward carries no autograd graph")
    if not model.training:
        # (not model.train() here: it would put the BatchNorm layers fit() froze with freeze_bn() back into training mode, main.py:405-406)
        raise RuntimeError("train_step needs the model in training mode: model.train(); model.freeze_bn() -- as fit() does per epoch")
    from . import kernels as K
    K.prefetch_amax(p for p in model.parameters() if p.dim() == 2)     # the step re-packs every weight: their maxima in one read-back
    out = model(sample)
    dev = out["disp"].device
    loss_dict = criterion(out, {"disp": sample["disp"].to(dev).clone(), "valid": sample["valid"].to(dev)})
    wd = criterion.weight_dict
    losses = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    for p in model.parameters():
        p.grad = None                                              # (main.py:419-421: "more efficient zero_grad")
    if reducer is not None:
        reducer.prepare()
    losses.backward()
    params = [p for g in optimizer.param_groups for p in g["params"]]
    if reducer is not None:
        reducer.finish()                                           # buckets reduced while backward ran (OverlappedGradientReducer)
    else:
        allreduce_gradients(params, group)
    torch.nn.utils.clip_grad_norm_(params, grad_clip)
    optimizer.step()
    return float(losses.detach()), {k: float(v.detach()) for k, v in loss_dict.items()}


def build_lr_scheduler(optimizer, cfg, last_step=-1):
    """The reference's schedule (main.py:380-388): OneCycleLR to BASE_LR over MAX_ITER + 100 steps, 5 % warm-up, cosine anneal, no momentum
    cycling; `last_step` = the resumed step count (a resumed optimizer state carries `initial_lr`, which OneCycleLR needs then)."""
    return torch.optim.lr_scheduler.OneCycleLR(optimizer, cfg.SOLVER.BASE_LR, cfg.SOLVER.MAX_ITER + 100, pct_start=0.05, cycle_momentum=False,
                                               anneal_strategy="cos", last_epoch=last_step if last_step > 0 else -1)


def save_checkpoint(path, model, optimizer=None, step=0, epoch=0):
    """The reference's two checkpoint layouts (main.py:441-458): {'model'} for `step_%06d.pth`, + {'optimizer', 'step', 'epoch'} for
    `checkpoint_latest.pth`.  State-dict keys are the reference's (strict-load contract of the model), so either side resumes the other's."""
    ckpt = {"model": model.state_dict()}
    if optimizer is not None:
        ckpt.update(optimizer=optimizer.state_dict(), step=int(step), epoch=int(epoch))
    torch.save(ckpt, path)


def load_checkpoint(path, model, optimizer=None, strict=True, map_location="cuda"):
    """Resume as main.py:352-372 does: weights from ckpt['model'] (or a bare state dict); optimizer, step and epoch when the file has all
    three and an optimizer is given (SOLVER.NO_RESUME_OPTIMIZER = pass optimizer=None).  Returns (epoch, step)."""
    ckpt = torch.load(path, map_location=map_location)
    model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt, strict=strict)
    if optimizer is not None and all(k in ckpt for k in ("optimizer", "step", "epoch")):
        optimizer.load_state_dict(ckpt["optimizer"])
        return int(ckpt["epoch"]), int(ckpt["step"])
    return 0, 0


def fit(model, criterion, optimizer, loader, cfg, checkpoint_dir=None, start_step=0, start_epoch=0, group=None, on_step=None,
        set_epoch=None):
    """The training loop of main.py:403-483 around train_step: model.train() + freeze_bn() per epoch, the OneCycle schedule stepped after
    every optimizer step, `step_%06d.pth` every SOLVER.CHECKPOINT_PERIOD steps and at the end, `checkpoint_latest.pth` (with optimizer,
    step, epoch) every SOLVER.LATEST_CHECKPOINT_PERIOD, both written by rank 0 only; stops at SOLVER.MAX_ITER.  `loader`: any iterable of
    samples {'img1', 'img2', 'disp', 'valid'}, re-iterated per epoch; set_epoch(epoch): the DistributedSampler hook of main.py:409-410;
    on_step(step, lr, total, loss_dict): the caller's logging (the reference writes TensorBoard scalars there).  Evaluation
    (TEST.EVAL_PERIOD) is the caller's: nmrf_amd.driver runs the inference path.  Returns (step, epoch)."""
    import os
    rank0 = not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0
    sched = build_lr_scheduler(optimizer, cfg, start_step)
    step, epoch, s = int(start_step), int(start_epoch), cfg.SOLVER
    reducer = OverlappedGradientReducer([p for g in optimizer.param_groups for p in g["params"]], group)
    if not reducer.active:
        reducer = None
    try:
        while step < s.MAX_ITER:
            model.train()
            model.freeze_bn()
            if set_epoch is not None:
                set_epoch(epoch)
            for sample in loader:
                total, loss_dict = train_step(model, criterion, optimizer, sample, s.GRAD_CLIP, group, reducer)
                lr = sched.get_last_lr()[0]
                sched.step()
                step += 1
                if on_step is not None:
                    on_step(step, lr, total, loss_dict)
                if checkpoint_dir is not None and rank0:
                    if step % s.CHECKPOINT_PERIOD == 0 or step == s.MAX_ITER:
                        save_checkpoint(os.path.join(checkpoint_dir, "step_%06d.pth" % step), model)
                    if step % s.LATEST_CHECKPOINT_PERIOD == 0:
                        save_checkpoint(os.path.join(checkpoint_dir, "checkpoint_latest.pth"), model, optimizer, step, epoch)
                if step >= s.MAX_ITER:
                    return step, epoch
            epoch += 1
        return step, epoch
    finally:
        if reducer is not None:
            reducer.close()
