"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header declares
(no compute without a GPU), config/padder/state-dict mirror the reference interface, the product
refuses to run without a GPU (no fallback), sharding arithmetic."""
import ctypes
import os
import re

import pytest
import torch

from nmrf_amd import _lib
from nmrf_amd.config import get_cfg
from nmrf_amd.frame_utils import InputPadder
from nmrf_amd.models import build_model
from nmrf_amd.parallel import shard_range
from tests.util import build_product

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built_library():
    if not os.path.exists(_lib.LIB_PATH):
        from nmrf_amd.build import build_library
        build_library(verbose=False)


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "nmrf_hip.h")).read()
    declared = set(re.findall(r"\b(nmrf_\w+)\s*\(", hdr))
    assert len(declared) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/nmrf_hip.h but not exported"
    assert declared - {"nmrf_strerror"} == set(_lib.PROTOTYPES), "ctypes binding out of sync with the header"
    lib = _lib.load()
    assert lib.nmrf_abi_version() == _lib.ABI_VERSION
    assert lib.nmrf_strerror(-1).decode().startswith("invalid")
    # the reference kernels of the tools / test build (round-1 fp32-MFMA linears, Winograd convolution, 32-token block kernel) are
    # NOT part of the product library: declared in include/nmrf_hip_debug.h, exported by libnmrf_hip_debug.so only
    dbg_hdr = open(os.path.join(ROOT, "include", "nmrf_hip_debug.h")).read()
    dbg_declared = set(re.findall(r"^int\s+(nmrf_\w+)\s*\(", dbg_hdr, re.M))
    assert dbg_declared == set(_lib.DEBUG_PROTOTYPES) and not (dbg_declared & declared)
    for name in dbg_declared:
        assert not hasattr(raw, name), f"{name} is a debug-library entry point but libnmrf_hip.so exports it"
    if os.path.exists(_lib.DEBUG_LIB_PATH):
        dbg = ctypes.CDLL(_lib.DEBUG_LIB_PATH)
        for name in dbg_declared | declared:
            assert hasattr(dbg, name), f"{name} missing from libnmrf_hip_debug.so"
        # a tools build left behind by an older ABI fails every A/B test on the GPU box (it travels with the tree as built)
        assert dbg.nmrf_abi_version() == _lib.ABI_VERSION, "libnmrf_hip_debug.so is stale: python -m nmrf_amd.build"


def test_build_stamp_ties_both_libraries_to_the_sources_in_the_tree():
    """nmrf_build_stamp() = "abi<N>-<hash of every source and header>": equal for the product and the tools library of one build and
    equal to the hash of the tree (tests/conftest.py refuses to start otherwise); a tools library from other sources is refused
    when it is loaded next to the product library."""
    from nmrf_amd import build
    want = "abi%d-%s" % (_lib.ABI_VERSION, build.source_stamp())
    assert _lib.load().nmrf_build_stamp().decode() == want
    assert re.fullmatch(r"abi\d+-[0-9a-f]{16}", want)
    if os.path.exists(_lib.DEBUG_LIB_PATH):
        assert build.library_stamp(_lib.DEBUG_LIB_PATH) == want
        assert _lib.load_debug().nmrf_build_stamp().decode() == want
    # the refusal itself, on a copy of the product library presented as the tools library of a differently stamped product
    import shutil, subprocess, sys, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fake = os.path.join(tmp, "libnmrf_hip_debug.so")
        shutil.copy(_lib.LIB_PATH, fake)
        code = ("import ctypes, nmrf_amd._lib as L\n"
                "L.DEBUG_LIB_PATH = %r\n"
                "L.DEBUG_PROTOTYPES = {}\n"
                "lib = L.load()\n"
                "real = lib.nmrf_build_stamp\n"
                "class Fake:\n"
                "    restype = None\n"
                "    def __call__(self): return b'abi0-0000000000000000'\n"
                "lib.nmrf_build_stamp = Fake()\n"
                "try:\n"
                "    L.load_debug()\n"
                "except L.NmrfHipError as e:\n"
                "    assert 'other sources' in str(e); print('refused')\n") % fake
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
        assert out.stdout.strip() == "refused", out.stdout + out.stderr


def test_product_reads_two_environment_switches_only():
    """NMRF_LINEAR (split | fp32: A/B runs on the debug library's fp32-MFMA linears) and NMRF_OVERLAP (side stream on / off for
    profiling) -- nothing else in the product tree changes the arithmetic or the launch structure from the environment."""
    names = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nmrf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                names |= set(re.findall(r"(?:environ(?:\.get)?\s*[\(\[]\s*|getenv\s*\(\s*)\"(\w+)\"", txt))
    # (HIPCC: the build script's compiler override; the two getenvs are tuning overrides of the debug build, under NMRF_DEBUG_PROBES)
    assert names <= {"NMRF_LINEAR", "NMRF_OVERLAP", "HIPCC", "NMRF_TL_NOPIPE", "NMRF_STRIPE_KSPLIT"}, names


def test_entry_points_validate_arguments_without_touching_the_gpu():
    lib = _lib.load()
    assert lib.nmrf_cost_volume_f32(None, None, 1, 256, 4, 4, 40, 4, None, None) == -3       # NMRF_ENULL
    one = ctypes.c_void_p(16)
    assert lib.nmrf_cost_volume_f32(one, one, 1, 255, 4, 4, 40, 4, one, None) == -1           # C % G != 0
    assert lib.nmrf_nms_topk_f32(one, 10, 300, 4, 1e-3, 1, one, None) == -1                   # D > 64
    assert lib.nmrf_window_attn_f32(one, one, 1, 13, 12, 4, 128, 4, 6, 0, 1, 0, one, None, None) == -1 # Hp % win
    assert lib.nmrf_stripe_attn_f32(one, one, one, 1, 4, 4, 4, 128, 0, 0, one, None, None) == -1    # axes == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnmrf_hip.so")
    with pytest.raises(_lib.NmrfHipError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_product_has_no_cpu_path():
    model = build_product(128)
    with pytest.raises(RuntimeError, match="no CPU"):
        model({"img1": torch.zeros(1, 3, 32, 64), "img2": torch.zeros(1, 3, 32, 64)})
    from nmrf_amd import kernels as K
    with pytest.raises(_lib.NmrfHipError):
        K.nms_topk(torch.rand(8, 40), 4, 1e-3)
    with pytest.warns(UserWarning, match="FORWARD only"), pytest.raises(RuntimeError, match="no CPU"):   # train(): forward-only, still no CPU path
        model.train()({"img1": torch.zeros(1, 3, 32, 64), "img2": torch.zeros(1, 3, 32, 64)})


def test_config_defaults_and_overrides():
    cfg = get_cfg()
    assert (cfg.DPN.MAX_DISP, cfg.DPN.COST_GROUP, cfg.DPN.NUM_PROPOSALS, cfg.DPN.CONTEXT_DIM) == (320, 4, 4, 64)
    assert (cfg.NMP.WINDOW_SIZE, cfg.NMP.REFINE_WINDOW_SIZE, cfg.NMP.SPLIT_SIZE) == (6, 4, 1)
    assert (cfg.NMP.NUM_PROP_LAYERS, cfg.NMP.NUM_INFER_LAYERS, cfg.NMP.NUM_REFINE_LAYERS) == (5, 5, 5)
    cfg.merge_from_list(["NMP.NUM_INFER_LAYERS", "4", "BACKBONE.COMPAT", "False"])
    assert cfg.NMP.NUM_INFER_LAYERS == 4 and cfg.BACKBONE.COMPAT is False
    with pytest.raises(KeyError):
        cfg.merge_from_list(["NMP.NOPE", 1])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SEED = 1
    c2 = cfg.clone()
    c2.defrost()
    c2.SEED = 7
    assert cfg.SEED == 326


def test_config_yaml_with_base(tmp_path):
    (tmp_path / "base.yaml").write_text("DATASETS:\n  DIVIS_BY: 32\nBACKBONE:\n  OUT_CHANNELS: 128\n")
    (tmp_path / "child.yaml").write_text("__BASE__: base.yaml\nBACKBONE:\n  COMPAT: False\n")
    cfg = get_cfg()
    cfg.merge_from_file(str(tmp_path / "child.yaml"))
    assert cfg.DATASETS.DIVIS_BY == 32 and cfg.BACKBONE.OUT_CHANNELS == 128 and cfg.BACKBONE.COMPAT is False


def test_input_padder_matches_reference_semantics():
    x = torch.arange(2 * 3 * 375 * 1242, dtype=torch.float32).view(2, 3, 375, 1242)
    p = InputPadder(x.shape, mode="proposal", divis_by=8)
    (y,) = p.pad(x)
    assert y.shape == (2, 3, 376, 1248)
    assert torch.equal(y[..., :375, :1242], x) and torch.equal(y[..., 375, :1242], x[..., 374, :])
    assert torch.equal(y[..., :375, 1242:], x[..., :, 1241:].expand(-1, -1, -1, 6))
    assert torch.equal(p.unpad(y), x)
    q = InputPadder((1, 3, 540, 960), mode="sintel", divis_by=32)
    assert q._pad == [0, 0, 2, 2]
    assert InputPadder((1, 3, 64, 64), mode="proposal")._pad == [0, 0, 0, 0]


def test_state_dict_contract():
    """Names/shapes the reference checkpoints carry; 351 tensors, 6 113 210 parameters."""
    model = build_model(get_cfg())[0]
    sd = model.state_dict()
    assert len(sd) == 351 and sum(p.numel() for p in model.parameters()) == 6113210
    expect = {
        "concatconv.0.weight": (128, 256, 3, 3), "gw.3.weight": (256, 128, 1, 1),
        "inference.ffn.fc1.weight": (128, 160), "inference.layers.0.self_nmp.q.weight": (128, 159),
        "inference.layers.4.nmp.qkv.weight": (384, 159),
        "inference.layers.1.nmp.attn.relative_position_enc_table": (121, 384),
        "inference.layers.1.nmp.attn.relative_position_index": (36, 36),
        "refinement.layers.0.nmp.attn.relative_position_enc_table": (49, 384),
        "refinement.layers.0.nmp.attn.relative_position_index": (16, 16),
        "infer_head.layers.2.weight": (64, 128), "infer_score_head.weight": (64, 128),
        "refine_head.layers.2.weight": (16, 128), "dpn.mlp.0.weight": (8, 4, 5), "dpn.mlp.4.weight": (1, 16, 5),
        "dpn.proj.3.weight": (64, 128, 1, 1), "dpn.propagation.cost_encoder.0.weight": (128, 36),
        "dpn.propagation.proj.weight": (128, 159), "dpn.propagation.layers.2.nmp.q.weight": (128, 192),
        "dpn.propagation.layers.2.nmp.attns.1.get_v.weight": (64, 1, 3, 3), "dpn.prop_head.layers.2.weight": (1, 128),
        "backbone.conv1.weight": (64, 3, 7, 7), "backbone.layer2.0.downsample.0.weight": (96, 64, 1, 1),
        "backbone.conv2.bias": (256,), "device_indicator_tensor": (0,),
    }
    for k, shp in expect.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert "dpn.propagation.proj.bias" not in sd
    cfg = get_cfg()
    cfg.BACKBONE.COMPAT = False
    assert any(k.startswith("image_encoder.") for k in build_model(cfg)[0].state_dict())
    # a state dict round-trips strictly
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)


def test_relative_position_index_formula():
    m = build_model(get_cfg())[0]
    idx = m.inference.layers[0].nmp.attn.relative_position_index
    w = 6
    for (i, j) in ((0, 0), (0, 35), (35, 0), (7, 20)):
        ai, bi, aj, bj = i // w, i % w, j // w, j % w
        assert int(idx[i, j]) == (ai - aj + w - 1) * (2 * w - 1) + (bi - bj + w - 1)


def test_fused_weight_cache_tracks_parameter_updates():
    """Packed weight streams are derived tensors: _FusedCache rebuilds them when (and only when) a source parameter changes."""
    from nmrf_amd.models.nmp import _FusedCache
    m = build_product(128)
    blk = m.inference.layers[0].self_nmp
    cache, calls = _FusedCache(), []

    def build():
        calls.append(1)
        return torch.cat((blk.q.weight, blk.k.weight), 0).clone()
    w1 = cache.get((blk.q.weight, blk.k.weight), build)
    assert cache.get((blk.q.weight, blk.k.weight), build) is w1 and len(calls) == 1
    with torch.no_grad():
        blk.q.weight.add_(1.0)
    w2 = cache.get((blk.q.weight, blk.k.weight), build)
    assert w2 is not w1 and len(calls) == 2 and torch.equal(w2[:128], blk.q.weight)


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_stem_space_to_depth_weight_is_the_same_convolution():
    """nmrf_amd.kernels.stem_s2d_weight: the 7x7 / stride-2 / pad-3 stem equals a 4x4 / stride-1 convolution (pad 2 before, 1
    after) over the 2x2 space-to-depth image with the transformed filter -- checked in fp64 on the CPU (the HIP path runs
    exactly this reformulation, tests/test_hip_kernels.py::test_stem_space_to_depth)."""
    import torch.nn.functional as F
    from nmrf_amd.kernels import stem_s2d_weight
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 24, 40, generator=g, dtype=torch.float64)
    w = torch.randn(8, 3, 7, 7, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, None, 2, 3)
    s2d = torch.zeros(2, 16, 12, 20, dtype=torch.float64)
    s2d[:, :12] = F.pixel_unshuffle(x, 2)
    got = F.conv2d(F.pad(s2d, (2, 1, 2, 1)), stem_s2d_weight(w))
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-12
    assert float(stem_s2d_weight(w)[:, 12:].abs().max()) == 0.0


def test_conv_plan_covers_the_backbone_channel_counts():
    """(strips, groups) of the direct conv: strips * groups * 32 == Co for every channel count of the CNN encoder and the heads;
    four strips only when the launch still fills the chip; unsupported counts fall back (None)."""
    from nmrf_amd.kernels import _conv3_plan
    for co in (64, 96, 128, 192, 256, 384):
        for tiles in (30, 240, 960, 5000):
            strips, groups = _conv3_plan(co, tiles)
            assert strips in (2, 3, 4) and strips * groups * 32 == co
    assert _conv3_plan(256, 240) == (4, 2) and _conv3_plan(256, 60) == (2, 4) and _conv3_plan(128, 240) == (2, 2)
    assert _conv3_plan(32, 100) is None and _conv3_plan(160, 100) is None


def test_kv16_format_restatement_against_a_scalar_reading_of_the_header():
    """kernels.to_kv16 (the torch statement of the kv16 row format that the GPU tests hold the block kernel to) against a
    value-by-value reading of include/nmrf_hip.h: k per 32-channel head as [32 hi halves | 32 lo halves], v as hi | lo << 16,
    q untouched; hi = rn_f16(x), lo = rn_f16(x - hi)."""
    import numpy as np
    import torch
    from nmrf_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(7, 384, generator=g) * torch.tensor([1e-3, 1.0, 300.0, 1.0, 1.0, 1.0, 1.0]).view(7, 1)
    out = K.to_kv16(qkv)
    assert torch.equal(out[:, :128], qkv[:, :128])
    raw = out.numpy().view(np.uint8).reshape(7, 384 * 4)
    x = qkv.numpy()
    for t in range(7):
        for c in (0, 1, 31, 32, 77, 127):
            hi = np.float16(x[t, 128 + c])
            lo = np.float16(x[t, 128 + c] - np.float32(hi))
            h, cc = divmod(c, 32)
            base = 128 * 4 + h * 128
            assert raw[t, base + 2 * cc: base + 2 * cc + 2].view(np.float16)[0] == hi
            assert raw[t, base + 64 + 2 * cc: base + 64 + 2 * cc + 2].view(np.float16)[0] == lo
            hv = np.float16(x[t, 256 + c])
            lv = np.float16(x[t, 256 + c] - np.float32(hv))
            word = raw[t, (256 + c) * 4: (256 + c) * 4 + 4]
            assert word[:2].view(np.float16)[0] == hv and word[2:].view(np.float16)[0] == lv
            assert abs(float(np.float32(hv) + np.float32(lv)) - float(x[t, 256 + c])) <= 2.0 ** -21 * abs(float(x[t, 256 + c])) + 2.0 ** -25     # (lo goes subnormal below |x| ~ 2^-3: absolute 2^-25, split_mfma.h)


def test_optimizer_groups_follow_the_reference_order_incl_sampling_offsets():
    """build_optimizer of main.py:186-245 gives modules named `*sampling_offsets*` (MSDeformAttn, Swin-T neck) BASE_LR x 0.1
    and lists its groups in a fixed order (plain, offsets, norms, trunk, trunk bias tables, enc tables): an optimizer state dict stores
    groups by position, so the layout must match for `checkpoint_latest.pth` to resume on either side.  Also: a parameter frozen by the
    caller stays frozen (main.py:206-207), and train_step does not flip BatchNorm back into training mode."""
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.train import build_slice_optimizer
    cfg = get_cfg()
    cfg.merge_from_list(["BACKBONE.MODEL_TYPE", "swin", "BACKBONE.OUT_CHANNELS", 128, "DATASETS.DIVIS_BY", 32, "SOLVER.WEIGHT_DECAY_NORM", 0.002,
                         "SOLVER.BACKBONE_LR_DECAY", 0.5])
    cfg.freeze()
    model = build_model(cfg)[0].train().enable_grad_slice(full=True)
    names = {id(p): k for k, p in model.named_parameters()}
    pinned = next(p for k, p in model.named_parameters() if k.endswith("infer_head.layers.0.weight"))
    pinned.requires_grad_(False)
    opt = build_slice_optimizer(model, cfg)
    assert not pinned.requires_grad and all(id(pinned) != id(p) for g in opt.param_groups for p in g["params"])
    kinds = []
    for g in opt.param_groups:
        ks = [names[id(p)] for p in g["params"]]
        if all("sampling_offsets" in k for k in ks):
            kinds.append("offsets")
            assert abs(g["lr"] - cfg.SOLVER.BASE_LR * 0.1) < 1e-15 and g["weight_decay"] == cfg.SOLVER.WEIGHT_DECAY and len(ks) >= 2
        elif all(k.startswith("image_encoder.backbone") for k in ks):
            kinds.append("trunk_tab" if all("relative_position_bias_table" in k for k in ks) else "trunk")
            assert abs(g["lr"] - cfg.SOLVER.BASE_LR * 0.5) < 1e-15
        elif all("relative_position_enc_table" in k for k in ks):
            kinds.append("enc_tab")
            assert g["weight_decay"] == 0.0
        elif g["weight_decay"] == 0.002:
            kinds.append("norms")
        else:
            kinds.append("plain")
            assert not any("sampling_offsets" in k for k in ks) and g["lr"] == cfg.SOLVER.BASE_LR
    # (the trunk groups key on the prefix `image_encoder.backbone`, main.py:213, which no module of NMRF carries -- the Swin trunk lives
    #  under `backbone.` -- so they are empty on both sides; the order of the others is the reference's)
    order = ["plain", "offsets", "norms", "trunk", "trunk_tab", "enc_tab"]
    assert kinds == [k for k in order if k in kinds] and {"plain", "offsets", "norms", "enc_tab"} <= set(kinds), kinds
    # the reference's own function on the same module tree (parameter names are the strict-load contract): identical group layout
    ref_main = os.path.join("/root/reference", "main.py")
    if os.path.exists(ref_main):
        import ast
        src = open(ref_main).read()
        fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "build_optimizer")
        scope = {"torch": torch}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), ref_main, "exec"), scope)
        pinned.requires_grad_(False)
        ref_opt = scope["build_optimizer"](model, cfg)
        assert len(ref_opt.param_groups) == len(opt.param_groups)
        for a, b in zip(ref_opt.param_groups, opt.param_groups):
            assert [id(p) for p in a["params"]] == [id(p) for p in b["params"]]
            assert abs(a["lr"] - b["lr"]) < 1e-15 and a["weight_decay"] == b["weight_decay"]
        opt.load_state_dict(ref_opt.state_dict())                     # a reference `checkpoint_latest.pth` resumes here


def test_training_host_side_optimizer_groups_schedule_and_checkpoints(tmp_path):
    """nmrf_amd.train, host side (no GPU): the optimizer groups of main.py:186-245 on the whole model (relative-position tables without
    weight decay, LayerNorm parameters at WEIGHT_DECAY_NORM), the OneCycle schedule of main.py:380-388, and the two checkpoint layouts of
    main.py:441-458 -- saved and resumed (weights, AdamW moments, step, epoch), loadable by the reference (same state-dict keys)."""
    import json
    from nmrf_amd.config import get_cfg
    from nmrf_amd.models import build_model
    from nmrf_amd.train import build_lr_scheduler, build_slice_optimizer, load_checkpoint, save_checkpoint, slice_parameters
    cfg = get_cfg()
    cfg.merge_from_list(["SOLVER.MAX_ITER", 50, "SOLVER.WEIGHT_DECAY_NORM", 0.002])
    cfg.freeze()
    model = build_model(cfg)[0].enable_training()
    assert model.training and model.grad_slice and model.grad_full and not build_model(cfg)[0].enable_training(convolutions=False).grad_full
    opt = build_slice_optimizer(model, cfg)
    names = {id(p): k for k, p in model.named_parameters()}
    assert sorted(names[id(p)] for g in opt.param_groups for p in g["params"]) == sorted(names.values())
    by = {names[id(p)]: g for g in opt.param_groups for p in g["params"]}
    assert by["inference.layers.0.nmp.attn.relative_position_enc_table"]["weight_decay"] == 0.0
    assert by["refinement.norm.weight"]["weight_decay"] == 0.002 and by["inference.layers.1.nmp.norm2.bias"]["weight_decay"] == 0.002
    assert by["backbone.conv1.weight"]["weight_decay"] == cfg.SOLVER.WEIGHT_DECAY and by["backbone.conv1.weight"]["lr"] == cfg.SOLVER.BASE_LR
    # the slice: the convolutional side frozen
    sl = build_model(cfg)[0].train().enable_grad_slice()
    build_slice_optimizer(sl, cfg)
    frozen = [k for k, p in sl.named_parameters() if not p.requires_grad]
    assert len(slice_parameters(sl)) == 315 and frozen and all(k.startswith(("backbone.", "concatconv.", "gw.", "dpn.proj.")) for k in frozen)
    # schedule: warm-up over 5 % of MAX_ITER + 100 steps to BASE_LR, cosine down
    sched = build_lr_scheduler(opt, cfg)
    lrs = []
    for _ in range(cfg.SOLVER.MAX_ITER):
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
    peak = max(range(len(lrs)), key=lrs.__getitem__)
    assert lrs[peak] > 0.99 * cfg.SOLVER.BASE_LR and 6 <= peak <= 8 and lrs[0] < 0.05 * cfg.SOLVER.BASE_LR and all(a > b for a, b in zip(lrs[peak:], lrs[peak + 1:]))
    # checkpoints
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01)
    save_checkpoint(str(tmp_path / "step.pth"), model)
    save_checkpoint(str(tmp_path / "latest.pth"), model, opt, step=37, epoch=3)
    assert set(torch.load(str(tmp_path / "step.pth"))) == {"model"}
    other = build_model(cfg)[0].train().enable_grad_slice(full=True)
    opt2 = build_slice_optimizer(other, cfg)
    assert load_checkpoint(str(tmp_path / "step.pth"), other, opt2, map_location="cpu") == (0, 0)
    assert load_checkpoint(str(tmp_path / "latest.pth"), other, opt2, map_location="cpu") == (3, 37)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), other.state_dict().values()))
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and all(torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"]) for k in s1)
    resumed = build_lr_scheduler(opt2, cfg, last_step=37)
    assert abs(resumed.get_last_lr()[0] - lrs[38]) < 1e-12          # (last_epoch = start_step as main.py:378-388: the constructor's own step makes it 38)
    with open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")) as f:
        assert list(json.load(f)["default"]) == list(torch.load(str(tmp_path / "step.pth"))["model"])


def test_integration_appendix_lists_every_entry_point():
    """INTEGRATION.md's appendix (tools/gen_abi_index.py) is generated from include/nmrf_hip.h: in step with the header, one row per
    exported symbol of the binding, and every A-row / N-row entry point cites reference lines."""
    import subprocess
    import sys
    from nmrf_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.call([sys.executable, os.path.join(root, "tools", "gen_abi_index.py"), "--check"]) == 0
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    body = doc[doc.index("<!-- abi-index:begin -->"):doc.index("<!-- abi-index:end -->")]
    rows = {l.split("`")[1]: l for l in body.splitlines() if l.startswith("| `nmrf_")}
    assert set(rows) == set(_lib.PROTOTYPES) | {"nmrf_strerror"} or set(rows) == set(_lib.PROTOTYPES)
    uncited = [n for n, l in rows.items() if l.rstrip().endswith("| – |")]
    helpers = ("nmrf_selftest_", "nmrf_strerror", "nmrf_abi_version", "nmrf_pack_", "nmrf_host_", "nmrf_sum_partials", "nmrf_colsum_",
               "nmrf_act_bwd", "nmrf_absmax", "nmrf_from_kv16", "nmrf_layernorm", "nmrf_instance_stats", "nmrf_prep_images_s2d_f32", "nmrf_nmp_block16_clock_records", "nmrf_build_stamp")
    assert all(n.startswith(helpers) for n in uncited), [n for n in uncited if not n.startswith(helpers)]


def test_gelu_fast_coefficients_meet_their_stated_error():
    """csrc/common.h:gelu_fast (round 6): v Phi(v) = max(v, 0) - |v| 2^(-t P(t)) / 2 with t = min(|v|, 7) and the degree-5 polynomial whose
    coefficients are in the header.  Emulated in fp32 (every operation rounded once, as the kernel's v_fma / v_mul / v_exp do) against
    fp64 erf over [-12, 12]: the header states 3.0e-7 (half an ulp at 4), the form it replaced had 4.7e-7."""
    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(ROOT, "nmrf_amd", "csrc", "common.h")).read()
    body = src[src.index("__device__ __forceinline__ float gelu_fast(float v) {"):]
    body = body[:body.index("#else")]
    co = [float(x) for x in re.findall(r"(-?\d\.\d+(?:e-?\d+)?)f", body)]
    # fminf(a, 7.0f), then c5, c4 (first fma), c3, c2, c1, c0, then the -0.5f / 0.f of the last line
    assert co[0] == 7.0 and len(co) >= 7, co
    c5, c4, c3, c2, c1, c0 = co[1:7]
    f32 = np.float32

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)

    v = np.concatenate([np.linspace(-12, 12, 1200001), np.random.default_rng(0).normal(size=400000) * 2]).astype(f32)
    t = np.minimum(np.abs(v), f32(7.0))
    p = fma(np.full_like(v, f32(c5)), t, np.full_like(v, f32(c4)))
    for c in (c3, c2, c1, c0):
        p = fma(p, t, np.full_like(v, f32(c)))
    e = np.exp2((-t.astype(np.float64) * p.astype(np.float64)).astype(f32).astype(np.float64)).astype(f32)
    got = fma((e.astype(np.float64) * t.astype(np.float64)).astype(f32), np.full_like(v, f32(-0.5)), np.maximum(v, f32(0)))
    want = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
    err = np.abs(got - want)
    assert err.max() <= 3.2e-7, (float(err.max()), float(v[err.argmax()]))
    assert np.all(got[v > 8] == v[v > 8]) and np.all(np.abs(got[v < -8]) < 1e-10)     # the tails: identity / zero
