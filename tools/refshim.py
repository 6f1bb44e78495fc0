"""Import shims that let the *reference* (`/root/reference`, read-only) be
imported in the build container, where several of its third-party dependencies
(timm, yacs, omegaconf, cv2, imageio, termcolor, the compiled MSDA extension)
are absent.  Used ONLY by tools/gen_golden.py to produce the fixtures under
tests/golden/.  Nothing here (and nothing of the reference) travels to the GPU
box or is imported by the product, the tests or the bench.

The stubs restate only the *interface* the reference touches:
  timm.models.layers / timm.layers : Mlp, DropPath, to_2tuple, trunc_normal_
  yacs.config.CfgNode              : attribute dict with clone/freeze/merge
  omegaconf.DictConfig             : isinstance target only
  cv2 / imageio / termcolor        : import-time names only
  MultiScaleDeformableAttention    : forward routed to the reference's own
                                     pure-PyTorch ms_deform_attn_core_pytorch
"""
import copy
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class _Mlp(nn.Module):
    """timm 0.9.16 Mlp: fc1 -> act -> drop1 -> norm -> fc2 -> drop2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not (self.training and self.drop_prob > 0)
        return x


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def _trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


class _CfgNode(dict):
    """Minimal yacs.config.CfgNode."""
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        self.__dict__[_CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = type(self)(v) if isinstance(v, dict) and not isinstance(v, _CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[_CfgNode.IMMUTABLE]:
            raise AttributeError("frozen")
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = type(self)()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        return new

    def _set_immutable(self, flag):
        self.__dict__[_CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, _CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def is_frozen(self):
        return self.__dict__[_CfgNode.IMMUTABLE]

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and k in self and isinstance(self[k], _CfgNode):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v

    def merge_from_list(self, lst):
        assert len(lst) % 2 == 0
        for key, val in zip(lst[0::2], lst[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            old = node[parts[-1]]
            if isinstance(val, str):
                import ast
                try:
                    val = ast.literal_eval(val)
                except Exception:
                    pass
            if old is not None and not isinstance(old, (list, tuple)) and val is not None:
                val = type(old)(val)
            node[parts[-1]] = val


def install(third_party_only=False):
    """third_party_only=True: only the stand-ins for timm / yacs / omegaconf / cv2 / imageio / termcolor (what a user's
    environment would provide) -- no MSDA stand-in and no sys.path entry: tests/test_dropin_cpu.py uses this to run the real
    checkout's import lines through nmrf_amd.dropin."""
    if "nmrf" in sys.modules and getattr(sys.modules["nmrf"], "__file__", "").startswith(REF):
        return
    timm = _mod("timm")
    tml = _mod("timm.models")
    tl = _mod("timm.models.layers")
    tl2 = _mod("timm.layers")
    for m in (tl, tl2):
        m.Mlp, m.DropPath, m.to_2tuple, m.trunc_normal_ = _Mlp, _DropPath, _to_2tuple, _trunc_normal_
    timm.models, timm.layers, tml.layers = tml, tl2, tl

    yacs = _mod("yacs")
    yc = _mod("yacs.config")
    yc.CfgNode = _CfgNode
    yacs.config = yc

    oc = _mod("omegaconf")
    oc.DictConfig = type("DictConfig", (), {})

    cv2 = _mod("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda f: None)
    _mod("imageio")
    tc = _mod("termcolor")
    tc.colored = lambda s, *a, **k: s

    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except ImportError:                                  # nmrf/utils/misc.py:23-24, nmrf/data/transforms.py:14: names only
            tv = _mod("torchvision")
            tv.__version__ = "0.20.0"
            tvt = _mod("torchvision.transforms")
            tvt.ColorJitter = tvt.Compose = type("_Unused", (), {"__init__": lambda self, *a, **k: None})
            tvt.functional = _mod("torchvision.transforms.functional")
            tv.transforms = tvt

    if third_party_only:
        return
    msda = _mod("MultiScaleDeformableAttention")

    def _fwd(value, shapes, lvl_start, loc, w, im2col_step):
        from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
        return ms_deform_attn_core_pytorch(value, shapes.tolist(), loc, w)

    msda.ms_deform_attn_forward = _fwd

    def _bwd(value, shapes, start, loc, w, grad_output, im2col_step):
        # the compiled extension's backward, restated as autograd of the reference's own pure-PyTorch formulation (fixtures of the
        # Swin configuration's training step: tools/gen_golden.py:run_train_swin)
        import torch
        from ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
        with torch.enable_grad():
            v, l, ww = (t.detach().clone().requires_grad_(True) for t in (value, loc, w))
            out = ms_deform_attn_core_pytorch(v, shapes.tolist(), l, ww)
            return torch.autograd.grad(out, (v, l, ww), grad_output)

    msda.ms_deform_attn_backward = _bwd

    if REF not in sys.path:
        sys.path.insert(0, REF)
    # never let the build's own drop-in alias packages shadow the reference
    for name in [n for n in sys.modules if n == "nmrf" or n.startswith("nmrf.") or n == "ops" or n.startswith("ops.")]:
        del sys.modules[name]


def build_reference_model(opts=()):
    install()
    from nmrf.config import get_cfg
    from nmrf.models import build_model
    cfg = get_cfg()
    cfg.merge_from_list(list(opts))
    cfg.freeze()
    model = build_model(cfg)[0]
    model.eval()
    return model, cfg
