"""`ops.modules.MSDeformAttn` (ops/modules/ms_deform_attn.py:28-130) on the HIP operator."""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .functions import MSDeformAttnFunction


class MSDeformAttn(nn.Module):
    """Multi-scale deformable attention.  Parameters: sampling_offsets, attention_weights, value_proj, output_proj."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        self.im2col_step = 64
        self.d_model, self.n_levels, self.n_heads, self.n_points, self.ratio = d_model, n_levels, n_heads, n_points, ratio
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, int(d_model * ratio))
        self.output_proj = nn.Linear(int(d_model * ratio), d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        ang = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        ring = torch.stack((ang.cos(), ang.sin()), -1)
        ring = ring / ring.abs().max(-1, keepdim=True)[0]
        ring = ring.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        ring = ring * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(ring.reshape(-1))
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        n, len_q, _ = query.shape
        _, len_in, _ = input_flatten.shape
        if not (input_flatten.is_cuda and torch.cuda.is_current_stream_capturing()):
            # (reads the device tensor back: the reference's check, ms_deform_attn.py:97; skipped while a hipGraph is being captured)
            assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == len_in
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(n, len_in, self.n_heads, int(self.ratio * self.d_model) // self.n_heads)
        off = self.sampling_offsets(query).view(n, len_q, self.n_heads, self.n_levels, self.n_points, 2)
        aw = F.softmax(self.attention_weights(query).view(n, len_q, self.n_heads, self.n_levels * self.n_points), -1)
        aw = aw.view(n, len_q, self.n_heads, self.n_levels, self.n_points)
        if reference_points.shape[-1] == 2:
            norm = torch.stack((input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]), -1)
            loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + off / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), aw.contiguous(), self.im2col_step)
        return self.output_proj(out)
