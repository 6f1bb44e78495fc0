"""N4, first slice: one optimisation step over the part of the model whose gradients this build can produce on the MI355X
(nmrf_amd.models.NMRF.enable_grad_slice: the WHOLE inference and refinement stages with their three heads, and -- behind the proposal
loss -- the WHOLE propagation stage with its head, and the seed filter: 315 of the model's 351 tensors; the rest is the encoder, the
matching heads and the DPN context convolutions), shaped like the reference's training loop (main.py:403-430):

    model.train(); loss_dict = criterion(model(sample), sample); losses = sum_k weight_dict[k] * loss_dict[k]
    param.grad = None; losses.backward(); clip_grad_norm_(GRAD_CLIP); optimizer.step()

and, with more than one rank, the gradient average DistributedDataParallel performs for the reference (main.py:334-339) as ONE
bucketed all-reduce over RCCL (`allreduce_gradients`: the slice's gradients are 0.27 M floats -- a single 1 MB bucket; one process per
GPU, weights replicated, batch sharded as in nmrf_amd.parallel).  Everything else of the model stays frozen: its kernels are forward-only."""
import torch
import torch.distributed as dist

# the WHOLE inference, refinement and propagation stages (attention backward kernels, round 5) and the four prediction heads
SLICE_PREFIXES = ("infer_head.", "infer_score_head.", "refine_head.", "inference.", "refinement.", "dpn.prop_head.", "dpn.propagation.",
                  "dpn.mlp.")                        # + the seed filter (Conv1d x 3 + softmax: the `init` loss)


def slice_parameters(model):
    """[(name, parameter)] the gradient slice reaches, in named_parameters() order."""
    return [(name, p) for name, p in model.named_parameters() if name.startswith(SLICE_PREFIXES)]


def build_slice_optimizer(model, cfg):
    """AdamW over the slice with the reference's grouping (main.py:186-245): plain parameters at BASE_LR / WEIGHT_DECAY, the LayerNorm
    parameters at WEIGHT_DECAY_NORM.  Every other parameter is frozen (requires_grad False): no kernel could fill its .grad."""
    keep = {id(p) for _, p in slice_parameters(model)}
    for p in model.parameters():
        p.requires_grad_(id(p) in keep)
    norm_ids = {id(p) for m in model.modules() if isinstance(m, torch.nn.LayerNorm) for p in m.parameters()}
    plain = [p for _, p in slice_parameters(model) if id(p) not in norm_ids]
    norms = [p for _, p in slice_parameters(model) if id(p) in norm_ids]
    groups = [{"params": plain, "lr": cfg.SOLVER.BASE_LR},
              {"params": norms, "lr": cfg.SOLVER.BASE_LR, "weight_decay": cfg.SOLVER.WEIGHT_DECAY_NORM}]
    return torch.optim.AdamW(groups, lr=cfg.SOLVER.BASE_LR, weight_decay=cfg.SOLVER.WEIGHT_DECAY)


def allreduce_gradients(params, group=None):
    """Average the gradients of `params` over the ranks of `group` with ONE all-reduce of a flat bucket (what DDP's reducer does
    bucket by bucket, main.py:334-339); identity without a process group.  A parameter without a gradient on this rank contributes
    zeros (DDP's find_unused_parameters semantics are not needed: the slice is the same on every rank)."""
    params = [p for p in params if p.requires_grad]
    if not params or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p).clone()
        off += n
    return flat.numel()


def train_step(model, criterion, optimizer, sample, grad_clip=1.0, group=None):
    """One step of main.py:413-430 on the gradient slice.  sample: {'img1', 'img2', 'disp', 'valid'} (H, W multiples of
    DATASETS.DIVIS_BY: the training-mode forward does not pad).  Returns (total loss as a float, the loss dict)."""
    if not getattr(model, "grad_slice", False):
        raise RuntimeError("call model.train().enable_grad_slice() first: without it the training-mode forward carries no autograd graph")
    model.train()
    out = model(sample)
    dev = out["disp"].device
    loss_dict = criterion(out, {"disp": sample["disp"].to(dev).clone(), "valid": sample["valid"].to(dev)})
    wd = criterion.weight_dict
    losses = sum(loss_dict[k] * wd[k] for k in loss_dict if k in wd)
    for p in model.parameters():
        p.grad = None                                              # (main.py:419-421: "more efficient zero_grad")
    losses.backward()
    params = [p for g in optimizer.param_groups for p in g["params"]]
    allreduce_gradients(params, group)
    torch.nn.utils.clip_grad_norm_(params, grad_clip)
    optimizer.step()
    return float(losses.detach()), {k: float(v.detach()) for k, v in loss_dict.items()}
