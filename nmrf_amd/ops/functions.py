"""`ops.functions` of the reference on the HIP MSDA kernels (ops/functions/ms_deform_attn_func.py:19-71)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import kernels as K


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """Extension entry point `MultiScaleDeformableAttention.ms_deform_attn_forward` (ops/src/vision.cpp:14).
    Same argument checks as ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:28-52); the im2col_step chunking
    of the reference is a launch detail of ITS kernel and is only validated here."""
    for name, t in (("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                    ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)):
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU" if name == "value" else "%s must be a CUDA tensor" % name)
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if batch % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    return K.msda_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """`MultiScaleDeformableAttention.ms_deform_attn_backward` (ops/src/vision.cpp:15) -> [grad_value, grad_loc, grad_w]."""
    batch = value.shape[0]
    step = min(batch, int(im2col_step))
    if batch % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (batch, step))
    return list(K.msda_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                grad_output.contiguous()))


class MSDeformAttnFunction(Function):
    """apply(value[N,S,M,D], shapes[L,2] i64, level_start[L] i64, loc[N,Lq,M,L,P,2], w[N,Lq,M,L,P], im2col_step)
    -> [N,Lq,M*D].  Half precision is cast up to fp32 like the reference's custom_fwd (ms_deform_attn_func.py:21)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        if value.dtype in (torch.float16, torch.bfloat16):
            value, sampling_locations, attention_weights = (t.float() for t in (value, sampling_locations,
                                                                                attention_weights))
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                        attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, w = ctx.saved_tensors
        gv, gl, gw = ms_deform_attn_backward(value, shapes, start, loc, w, grad_output.to(value.dtype), ctx.im2col_step)
        return gv, None, None, gl, gw, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """The reference's public debug helper (ms_deform_attn_func.py:49-71), kept for API parity: a grid_sample
    formulation of the same operator.  Never called by the product path."""
    n, _, m, d = value.shape
    _, lq, _, nl, p, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in value_spatial_shapes]
    grids = 2 * sampling_locations - 1
    out = value.new_zeros(n * m, d, lq)
    start = 0
    for lvl, (h, w) in enumerate(shapes):
        v = value[:, start:start + h * w].permute(0, 2, 3, 1).reshape(n * m, d, h, w)
        start += h * w
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(n * m, lq, p, 2)
        s = F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False)     # [NM,D,Lq,P]
        aw = attention_weights[:, :, :, lvl].permute(0, 2, 1, 3).reshape(n * m, 1, lq, p)
        out = out + (s * aw).sum(-1)
    return out.view(n, m * d, lq).transpose(1, 2).contiguous()
