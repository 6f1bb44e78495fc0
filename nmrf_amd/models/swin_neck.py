"""Swin-T image encoder + deformable-attention neck (`BACKBONE.MODEL_TYPE: swin`, configs/sceneflow_swint.yaml).

Stock PyTorch-ROCm by north_star (the backbone is out of hot-path scope) EXCEPT the multi-scale deformable
attention inside the neck, which runs on the HIP operator (nmrf_amd.ops.MSDeformAttn -> nmrf_msda_forward_f32).
Module tree and parameter names follow nmrf/models/swin.py, nmrf/models/adaptor_modules.py and
nmrf/models/backbone.py:101-158 so that `image_encoder.*` checkpoints load strictly.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..ops.modules import MSDeformAttn


# ------------------------------------------------------------------------------------------------
# Swin transformer (swin.py).  Tokens are kept as [B, H, W, C] between blocks.
# ------------------------------------------------------------------------------------------------
class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


def _windows(x, ws):
    """[B,Hp,Wp,C] -> [B*nW, ws*ws, C]"""
    b, hp, wp, c = x.shape
    x = x.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws * ws, c)


def _unwindows(w, ws, b, hp, wp):
    c = w.shape[-1]
    return w.view(b, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, c)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.num_heads, self.ws = num_heads, window_size
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        ys, xs = torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing="ij")
        co = torch.stack((ys.reshape(-1), xs.reshape(-1)))
        rel = co[:, :, None] - co[:, None, :]
        self.register_buffer("relative_position_index",
                             (rel[0] + window_size - 1) * (2 * window_size - 1) + rel[1] + window_size - 1)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask):
        bw, n, c = x.shape
        qkv = self.qkv(x).view(bw, n, 3, self.num_heads, c // self.num_heads).permute(2, 0, 3, 1, 4)
        attn = (qkv[0] * self.scale) @ qkv[1].transpose(-1, -2)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(n, n, -1)
        attn = attn + bias.permute(2, 0, 1)[None]
        if mask is not None:
            nw = mask.shape[0]
            attn = (attn.view(bw // nw, nw, self.num_heads, n, n) + mask[None, :, None]).view(bw, self.num_heads, n, n)
        out = (attn.softmax(-1) @ qkv[2]).transpose(1, 2).reshape(bw, n, c)
        return self.proj(out)


def _drop_path(x, rate, training):
    """Stochastic depth per sample (timm DropPath as swin.py:228-233,300-301 uses it): in training mode a residual branch is dropped for a
    whole sample with probability `rate` and the kept ones are scaled by 1 / (1 - rate); the identity in eval mode."""
    if rate == 0.0 or not training:
        return x
    keep = 1.0 - rate
    mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
    return x * (mask / keep)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, drop_path=0.0):
        super().__init__()
        self.ws, self.shift, self.drop_path = window_size, shift_size, float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, mask):
        """x [B,H,W,C]"""
        b, h, w, c = x.shape
        ws = self.ws
        y = F.pad(self.norm1(x), (0, 0, 0, (-w) % ws, 0, (-h) % ws))
        hp, wp = y.shape[1:3]
        if self.shift:
            y = torch.roll(y, (-self.shift, -self.shift), (1, 2))
        y = _unwindows(self.attn(_windows(y, ws), mask if self.shift else None), ws, b, hp, wp)
        if self.shift:
            y = torch.roll(y, (self.shift, self.shift), (1, 2))
        x = x + _drop_path(y[:, :h, :w], self.drop_path, self.training)
        return x + _drop_path(self.mlp(self.norm2(x)), self.drop_path, self.training)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        b, h, w, c = x.shape
        x = F.pad(x, (0, 0, 0, w % 2, 0, h % 2))
        x = torch.cat((x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]), -1)
        return self.reduction(self.norm(x))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, downsample, drop_path=None):
        super().__init__()
        self.ws, self.shift = window_size, window_size // 2
        self.blocks = nn.ModuleList(
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 drop_path[i] if drop_path else 0.0)
            for i in range(depth))
        self.downsample = PatchMerging(dim) if downsample else None

    def _shift_mask(self, h, w, device):
        ws, sh = self.ws, self.shift
        hp, wp = -(-h // ws) * ws, -(-w // ws) * ws
        ids = torch.zeros(1, hp, wp, 1, device=device)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
                ids[:, hs, wsl] = cnt
                cnt += 1
        m = _windows(ids, ws).squeeze(-1)
        d = m[:, None, :] - m[:, :, None]
        return torch.where(d != 0, torch.full_like(d, -100.0), torch.zeros_like(d))      # swin.py:443-445 uses -100

    def forward(self, x):
        mask = self._shift_mask(x.shape[1], x.shape[2], x.device)
        for blk in self.blocks:
            x = blk(x, mask)
        return x, (self.downsample(x) if self.downsample is not None else x)


class PatchEmbed(nn.Module):
    def __init__(self, patch=4, cin=3, dim=96):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(cin, dim, patch, patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        h, w = x.shape[-2:]
        x = F.pad(x, (0, (-w) % self.patch, 0, (-h) % self.patch))
        return self.norm(self.proj(x).permute(0, 2, 3, 1))           # [B,H/4,W/4,C]


class SwinTransformer(nn.Module):
    def __init__(self, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, drop_path_rate=0.0):
        super().__init__()
        self.patch_embed = PatchEmbed(4, 3, embed_dim)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]      # stochastic depth decay rule (swin.py:592-609)
        self.layers = nn.ModuleList(
            BasicLayer(embed_dim * 2 ** i, depths[i], num_heads[i], window_size, mlp_ratio, i < len(depths) - 1,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])])
            for i in range(len(depths)))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        """-> [p0 (1/4), p1 (1/8), p2 (1/16), p3 (1/32)] as NCHW maps"""
        x = self.patch_embed(x)
        outs = []
        for layer in self.layers:
            out, x = layer(x)
            outs.append(out.permute(0, 3, 1, 2).contiguous())
        return outs


# ------------------------------------------------------------------------------------------------
# Deformable-attention neck (adaptor_modules.py)
# ------------------------------------------------------------------------------------------------
class _DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)

    def forward(self, x, h, w):
        b, n, c = x.shape
        return self.dwconv(x.transpose(1, 2).reshape(b, c, h, w)).flatten(2).transpose(1, 2)


class ConvFFN(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.dwconv = _DWConv(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x, h, w):
        return self.fc2(F.gelu(self.dwconv(self.fc1(x), h, w)))


class Extractor(nn.Module):
    def __init__(self, dim, num_heads=8, n_points=4, deform_ratio=1.0, cffn_ratio=0.25):
        super().__init__()
        self.query_norm = nn.LayerNorm(dim, eps=1e-6)
        self.feat_norm = nn.LayerNorm(dim, eps=1e-6)
        self.attn = MSDeformAttn(d_model=dim, n_levels=1, n_heads=num_heads, n_points=n_points, ratio=deform_ratio)
        self.ffn = ConvFFN(dim, int(dim * cffn_ratio))
        self.ffn_norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, query, reference_points, feat, spatial_shapes, level_start_index, h, w):
        query = query + self.attn(self.query_norm(query), reference_points, self.feat_norm(feat), spatial_shapes,
                                  level_start_index, None)
        return query + self.ffn(self.ffn_norm(query), h, w)


class ConvStem(nn.Module):
    def __init__(self, inplanes=64, out_channels=256):
        super().__init__()
        n = nn.InstanceNorm2d
        self.stem = nn.Sequential(
            nn.Conv2d(3, inplanes, 3, 2, 1, bias=False), n(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, 3, 1, 1, bias=False), n(inplanes), nn.ReLU(inplace=True),
            nn.Conv2d(inplanes, inplanes, 3, 1, 1, bias=False), n(inplanes), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1))
        self.fc = nn.Conv2d(inplanes, out_channels, 1)

    def forward(self, x):
        return self.fc(self.stem(x)).flatten(2).transpose(1, 2)       # [B, HW/16, C]


class DeformNeck(nn.Module):
    def __init__(self, dim, in_channel_list, num_heads=8, n_points=4, cffn_ratio=0.25, deform_ratio=1.0):
        super().__init__()
        self.dim = dim
        self.stem = ConvStem(64, dim)
        self.extractors = nn.ModuleList(Extractor(dim, num_heads, n_points, deform_ratio, cffn_ratio) for _ in range(4))
        self.fcs = nn.ModuleList(nn.Sequential(nn.LayerNorm(c, eps=1e-6), nn.Linear(c, dim)) for c in in_channel_list)

    def forward(self, image, features):
        b, _, hh, ww = image.shape
        h, w = hh // 4, ww // 4
        dev = image.device
        # level shapes are built on the host and copied once per (size, device): a pageable H2D copy inside forward would
        # synchronise every call and cannot be captured into a hipGraph
        key = (hh, ww, str(dev))
        if not hasattr(self, "_geom"):
            self._geom = {}
        if key not in self._geom:
            shp = torch.as_tensor([(hh // s, ww // s) for s in (4, 8, 16, 32)], dtype=torch.long, device=dev)
            self._geom[key] = (shp, shp.new_zeros((1,)))
        shapes, start = self._geom[key]
        ry = (torch.arange(h, dtype=torch.float32, device=dev) + 0.5) / h
        rx = (torch.arange(w, dtype=torch.float32, device=dev) + 0.5) / w
        ref = torch.stack((rx[None, :].expand(h, w), ry[:, None].expand(h, w)), -1).reshape(1, h * w, 1, 2)
        c = self.stem(image)
        for i, feat in enumerate(features):
            f = self.fcs[i](feat.flatten(2).transpose(1, 2))
            c = self.extractors[i](c, ref, f, shapes[i:i + 1].contiguous(), start, h, w)
        return c.transpose(1, 2).reshape(b, self.dim, h, w)


class SwinAdaptor(nn.Module):
    def __init__(self, out_channels, drop_path_rate=0.0):
        super().__init__()
        self.backbone = SwinTransformer(drop_path_rate=drop_path_rate)         # (the neck's own drop_path is 0: backbone.py:111-117)
        self.neck = DeformNeck(out_channels, [96, 192, 384, 768], deform_ratio=0.5)
        self.output_dim = out_channels
        self.register_buffer("mean", torch.tensor([123.675, 116.28, 103.53]).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([58.395, 57.12, 57.375]).view(1, 3, 1, 1))

    def forward(self, x):
        x = (x - self.mean) / self.std
        out = self.neck(x, self.backbone(x))
        return [out, F.avg_pool2d(out, 2, 2)]
