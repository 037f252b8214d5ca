"""Three-stage Winograd F(4,5) x bf16x3 pipeline (msk_conv_wbf.hip: wbf_tin_k -> wbf_gemm_k -> wbf_tout_k) against
the float64 oracle, through the C ABI.

The matrix stage multiplies fp32 operands that were split EXACTLY into three bf16 pieces with six bf16 products per
fp32 product (fp32 accumulate); the claim under test is fp32-class accuracy: the same tolerance as every other
convolution kernel of this repo (tests/test_gpu_ops.py: 8e-6 * sqrt(K/1000 + 1) of max|ref|), and in addition an
error no larger than 1.5x that of the exact-fp32 Winograd kernels it replaces on the same inputs."""
import numpy as np
import pytest

import ctypes as C

from helpers import dev, rel_err, t_empty, t_from_ncdhw, t_to_ncdhw, vec, vec_back, vp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _operand_split():
    """The tests of this file address the exact bf16x3 pipeline ("conv_split" 3) unless they select the fp16 two-piece
    format themselves; every test leaves the context in the product configuration (conv_split 2)."""
    dev().set_option("conv_split", 3)
    yield
    dev().set_option("conv_split", 2)
    dev().set_option("bwd_fuse", -1)

from oracle import vnet_numpy as O  # noqa: E402


def _desc(k, s, p):
    from medicalseg_amd._lib import MskConvDesc
    return MskConvDesc(*k, *s, *p)


def _conv_tol(K):
    return 8e-6 * np.sqrt(K / 1000.0 + 1.0)


WBF_CASES = [
    # (Cin, Cout, (N, D, H, W))                       tile variant / what it exercises
    (32, 32, (2, 16, 32, 16)),      # CN 32: 16 x 32 tile, exact fit, T = 4
    (32, 32, (1, 30, 60, 8)),       # ragged d and h tiles
    (64, 32, (1, 15, 30, 12)),      # 4 chunks, CN 32
    (32, 64, (2, 16, 16, 8)),       # CN 64: 16 x 16 tile, two column fragments per workgroup
    (64, 64, (1, 20, 13, 16)),      # eligible only with the transform along D (tile roles (13, 16))
    (128, 128, (1, 8, 16, 8)),      # CN 128: 8 x 16 tile, split-K over the chunks
    (64, 256, (1, 15, 16, 4)),      # two column groups, T = 1
    (256, 128, (2, 8, 8, 8)),       # 8 x 8 tile (MR = 2), 16 chunks split
    (128, 128, (1, 16, 8, 7)),      # W % 4 != 0: transform along H, tile roles (7, 16)
    (64, 64, (1, 12, 16, 15)),      # transform along D
    (32, 32, (1, 32, 32, 9)),       # RAGGED transform axis (round 4): 9 -> three W tiles, the last with one output inside the volume
    (64, 64, (2, 16, 16, 6)),       # ragged, T = 2, the last tile half inside
    (256, 256, (1, 16, 16, 2)),     # the MRI bottom level's shape class: T = 1, two of four outputs exist (2x padding, second-try limit)
    (32, 64, (1, 9, 24, 16)),       # ragged along D: the transform axis is a strided one
]


# operand formats: 3 = exact bf16 x 3 pieces (A/B), 2 = fp16 x 2 pieces with tensor scales -- the PRODUCT default (round-3
# verdict, weakness 3: the tile-variant matrix ran with the non-product format only)
SPLIT_TAGS = {3: ("wbf_gemm_k", "wbf_wgrad_k"), 2: ("wbf_gemm_h2_k", "wbf_wgrad_h2_k")}


@pytest.mark.parametrize("split", [2, 3])
@pytest.mark.parametrize("case", WBF_CASES)
def test_wbf_fwd_dgrad_match_oracle(case, split):
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", split)
    rng = np.random.default_rng(cin * 11 + cout + D)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    dxt = t_empty(N, cin, D, H, W, fill=3.0)
    wp, bp = vec(w.ravel()), vec(b)
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    d.set_option("poison_scratch", 0xFF)   # NaN-poisoned scratch: any read of an unwritten V / M slot shows
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        e_f, e_d = rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref)
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
        e_acc = rel_err(t_to_ncdhw(dxt), 2 * dx_ref)
        d.prof_enable(False)
        rep = d.prof_report()
        assert rep.get(SPLIT_TAGS[split][0], (0, 0))[0] >= 1, rep          # the pipeline really ran (the forward at least;
        # the data gradient swaps the channel roles and may pick another tile class or the fp32 kernels)
        # the exact-fp32 Winograd / direct kernels on the same inputs
        d.set_option("wino_bf3", 0)
        y2, dx2 = t_empty(N, cout, D, H, W, fill=7.0), t_empty(N, cin, D, H, W, fill=3.0)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), y2.msk())
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dx2.msk(), 0)
        o_f, o_d = rel_err(t_to_ncdhw(y2), y_ref), rel_err(t_to_ncdhw(dx2), dx_ref)
    finally:
        d.set_option("wino_bf3", 1)
        d.set_option("poison_scratch", -1)
    print(f"\nwbf {case} split {split}: fwd {e_f:.2e} (fp32 kernels {o_f:.2e})  dgrad {e_d:.2e} ({o_d:.2e})  acc {e_acc:.2e}")
    assert e_f < _conv_tol(cin * 125) and e_d < _conv_tol(cout * 125) and e_acc < _conv_tol(cout * 125)
    # fp32 class: within a small factor of the exact-fp32 kernels (direct MFMA ~5e-7, Winograd F(4,5) ~1.5e-6)
    assert e_f < 4e-6 and e_d < 4e-6


def test_wbf_channel_slices_and_fused_activation():
    """ld > c on both sides (the zero-copy concat slices of UpTransition) and the inference epilogue
    (msk_conv3d_fwd_act: bias + PReLU in wbf_tout_k)."""
    cin, cout, (N, D, H, W) = 32, 32, (1, 16, 28, 12)
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    slope = rng.uniform(0.05, 0.5, cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    act_ref = np.where(y_ref > 0, y_ref, y_ref * slope.reshape(1, -1, 1, 1, 1))
    xt = t_from_ncdhw(x, ld=48)
    yt = t_empty(N, cout, D, H, W, ld=40, fill=9.0)
    wp, bp, sp = vec(w.ravel()), vec(b), vec(slope)
    d.prof_reset()
    d.prof_enable(True)
    d.call("msk_conv3d_fwd_act", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), vp(sp), yt.msk())
    d.prof_enable(False)
    assert d.prof_report().get("wbf_gemm_k", (0, 0))[0] == 1
    got = yt.numpy()
    assert rel_err(got, act_ref) < _conv_tol(cin * 125)
    full = d.d2h(yt.ptr, (N, D, H, W, 40), np.float32)
    assert np.all(full[..., cout:] == 9.0)          # the slice's neighbours are untouched


WGRAD_CASES = [
    # (Cin, Cout, (N, D, H, W))
    (32, 32, (2, 16, 32, 16)),      # 8 x 16 tiles, many tiles per workgroup (split K over tiles)
    (32, 64, (1, 14, 30, 8)),       # ragged tiles, two output-channel blocks
    (64, 32, (2, 8, 8, 8)),         # 8 x 8 tiles (TH = 8), 4 input chunks
    (128, 128, (1, 8, 16, 4)),      # T = 1
    (32, 32, (1, 16, 8, 7)),        # W % 4 != 0: transform along another axis
    (64, 64, (1, 12, 16, 15)),      # transform along D
    (32, 32, (1, 32, 32, 9)),       # ragged transform axis: A dy of a tile's missing outputs is zero
    (64, 64, (2, 16, 16, 6)),
    (256, 256, (1, 16, 16, 2)),
    (128, 128, (1, 8, 16, 7)),      # round 5: the row-coalesced reduce (>= 128 channels) with a permuted transform axis
]


@pytest.mark.parametrize("split", [2, 3])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wbf_wgrad_matches_oracle(case, split):
    """wbf_wgrad_k (transposing LDS reads, v_mfma_f32_16x16x32_bf16, six products per fp32 product) + its split-K / G^T
    reduce against the float64 oracle, fresh and accumulating, and next to the exact-fp32 kernels on the same inputs."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", split)
    rng = np.random.default_rng(cin * 13 + cout + H)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    dw_ref, db_ref = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    nw = cout * cin * 125
    dwp, dbp = vec(np.full(nw, 0.5, np.float32)), vec(np.zeros(cout, np.float32))
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    d.set_option("wgrad_async", 0)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        got = d.d2h(dwp, (nw,), np.float32).reshape(dw_ref.shape)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 1)
        got2 = d.d2h(dwp, (nw,), np.float32).reshape(dw_ref.shape)
        d.prof_enable(False)
        rep = d.prof_report()
        assert rep.get(SPLIT_TAGS[split][1], (0, 0))[0] == 2, rep
        d.set_option("wino_bf3", 0)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        old = d.d2h(dwp, (nw,), np.float32).reshape(dw_ref.shape)
    finally:
        d.set_option("wino_bf3", 1)
        d.set_option("wgrad_async", 1)
    e, e2, eo = rel_err(got, dw_ref), rel_err(got2, 2 * dw_ref), rel_err(old, dw_ref)
    M = N * D * H * W
    print(f"\nwbf wgrad {case} split {split}: {e:.2e} (accumulating {e2:.2e}; fp32 kernels {eo:.2e})")
    assert e < _conv_tol(M) and e2 < _conv_tol(M)
    assert e < 4e-6


@pytest.mark.parametrize("case", [(32, 32, (2, 16, 32, 16)), (64, 128, (1, 8, 16, 8)), (32, 64, (1, 14, 30, 8)),
                                  (32, 32, (1, 32, 32, 9)), (64, 64, (2, 16, 16, 6)),   # ragged transform axis: statistics count what exists
                                  (16, 16, (1, 8, 8, 8)), (1, 16, (2, 9, 20, 40))])   # last: in_tr.conv1, statistics in conv_c1_mfma_k's epilogue (ragged tiles)
def test_conv3d_fwd_ex_stats_and_kept_transform(case):
    """msk_conv3d_fwd_ex: (a) the BatchNorm statistics record taken in the output transform equals msk_bn_stats of the
    stored y (float64 oracle: mean / M2 of y_ref); (b) the transformed input it leaves in the caller's buffer gives the
    SAME weight gradient bits through msk_conv3d_wgrad_ex as the transform msk_conv3d_wgrad recomputes; the last case is
    not eligible for the pipeline (16 channels): xform_bytes = 0 and the statistics come from msk_bn_stats."""
    import ctypes as C
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(cin + cout + D)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    wp, bp = vec(w.ravel()), vec(b)
    stats = vec(np.zeros(2 * cout, np.float32))
    nbytes = int(d.lib.msk_conv3d_xform_bytes(d.ctx, _desc(k, s_, p), xt.msk(), cout))
    assert (nbytes > 0) == (cin >= 32)
    xf = d.malloc(nbytes) if nbytes else None
    d.set_option("wgrad_async", 0)
    try:
        d.call("msk_conv3d_fwd_ex", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk(), vp(stats), vp(xf))
        got = t_to_ncdhw(yt)
        assert rel_err(got, y_ref) < _conv_tol(cin * 125)
        st = d.d2h(stats, (2 * cout,), np.float32)
        yc = np.moveaxis(y_ref, 1, -1).reshape(-1, cout)
        mean_ref, m2_ref = yc.mean(0), ((yc - yc.mean(0)) ** 2).sum(0)
        assert np.abs(st[:cout] - mean_ref).max() < 1e-5 * (np.abs(mean_ref).max() + 1)
        assert np.abs(st[cout:] - m2_ref).max() < 1e-5 * m2_ref.max()
        nw = cout * cin * 125
        dw1, dw2 = vec(np.zeros(nw, np.float32)), vec(np.zeros(nw, np.float32))
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dw1), None, 0)
        d.call("msk_conv3d_wgrad_ex", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dw2), None, 0, vp(xf))
        a1, a2 = d.d2h(dw1, (nw,), np.float32), d.d2h(dw2, (nw,), np.float32)
        assert np.array_equal(a1, a2)
    finally:
        d.set_option("wgrad_async", 1)


K3_CASES = [
    # (Cin, Cout, (N, D, H, W))
    (32, 32, (2, 16, 16, 16)),      # CN 32: 16 x 16 tiles
    (64, 32, (1, 16, 32, 8)),
    (32, 64, (1, 8, 16, 12)),       # CN 64: 8 x 16 tiles
    (128, 128, (1, 8, 8, 8)),       # CN 128: 8 x 8 tiles, split K
]


@pytest.mark.parametrize("dy_mag", [1.0, 2e-7])
@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("case", K3_CASES)
def test_wbf_3x3x3_pipeline(case, fp16, dy_mag):
    """The same three-stage pipeline for 3 x 3 x 3 convolutions (Winograd F(4,3), 6 points; UNet3D's DoubleConvs):
    forward, data gradient and weight gradient against the float64 oracle,
      fp16 = 0: exact bf16x3 operands -> the fp32-class tolerance of every other convolution kernel;
      fp16 = 1: option "conv_fp16" (the fp16 matrix path of BASELINE configs[3]): fp16 operands in the Winograd domain,
                fp32 accumulate.  STATED fp16 TOLERANCE: 3e-3 of max|ref| (operand rounding 2^-11 ~ 4.9e-4 per value,
                amplified ~3x by the output transform; measured 1.0e-3 .. 1.8e-3).
      dy_mag = 2e-7: the per-voxel loss gradient of the full-size 2 x 192 x 192 x 64 configuration -- inside fp16's SUBNORMAL
                range; every fp16 operand tensor is scaled by a device-side power of two from its maximum (round-2 advisor
                finding: the single-fp16 form converted gradients unscaled and kept 0-3 bits of them)."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    d = dev()
    rng = np.random.default_rng(cin * 3 + cout + D + fp16)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dy = (rng.standard_normal(y_ref.shape) * dy_mag).astype(np.float32)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    dw_ref, _ = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    dxt = t_empty(N, cin, D, H, W, fill=3.0)
    wp, bp = vec(w.ravel()), vec(b)
    nw = cout * cin * 27
    dwp = vec(np.zeros(nw, np.float32))
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    d.set_option("wgrad_async", 0)
    d.set_option("conv_fp16", fp16)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), None, 0)
        e_f, e_d = rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref)
        e_w = rel_err(d.d2h(dwp, (nw,), np.float32).reshape(dw_ref.shape), dw_ref)
        d.prof_enable(False)
        rep = d.prof_report()
        gk, wk = ("wbf_gemm_f16_k", "wbf_wgrad_f16_k") if fp16 else ("wbf_gemm_k", "wbf_wgrad_k")
        assert rep.get(gk, (0, 0))[0] >= 1 and rep.get(wk, (0, 0))[0] == 1, rep
    finally:
        d.set_option("conv_fp16", 0)
        d.set_option("wgrad_async", 1)
    print(f"\nwbf 3^3 {case} fp16={fp16} |dy|~{dy_mag:g}: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e}")
    if fp16:
        assert e_f < 3e-3 and e_d < 3e-3 and e_w < 3e-3
    else:
        assert e_f < _conv_tol(cin * 27) and e_d < _conv_tol(cout * 27) and e_w < _conv_tol(N * D * H * W) * 2


@pytest.mark.parametrize("split", [3, 2])
@pytest.mark.parametrize("case", [(32, 5, (2, 16, 32, 16)), (64, 5, (1, 8, 16, 16)), (128, 5, (1, 8, 16, 8)), (32, 3, (2, 16, 16, 16)),
                                  (32, 5, (1, 30, 60, 8)), (32, 5, (1, 32, 32, 9)), (64, 5, (2, 16, 16, 6)), (16, 5, (1, 8, 8, 8))])
def test_conv3d_bwd_bnact_fused_equals_three_call_form(case, split):
    """msk_conv3d_bwd_bnact (backward of a LUConv unit, vnet.py:36-41): the fused form -- BatchNorm/PReLU backward evaluated
    inside wbf_tin_dual_k, which writes both transforms of dy; dy never stored -- against (a) the same entry point without
    `ybuf` (msk_affine_act_bwd_apply -> msk_conv3d_dgrad -> msk_conv3d_wgrad_ex through HBM) and (b) the float64 oracle of
    the three operations.  The last case (16 channels) is not eligible: bytes() = 0 and the call takes the three-call form.
    split = 2: the fp16 two-piece operand format ("conv_split"); dout is scaled to 1e-6 (real gradient magnitudes: far below
    fp16's normal range) so that the device-side power-of-two scaling from the reduce pass's maxima is what makes it work."""
    import ctypes as C
    c, K, (N, D, H, W) = case
    k, s_, p = (K,) * 3, (1, 1, 1), (K // 2,) * 3
    d = dev()
    rng = np.random.default_rng(c + K + D)
    f8 = lambda a: a.astype(np.float64)
    x = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((c, c) + k) / np.sqrt(c * K ** 3)).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    dout = (rng.standard_normal((N, c, D, H, W)) * (1e-6 if split == 2 else 1.0)).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, c).astype(np.float32), (0.3 * rng.standard_normal(c)).astype(np.float32)
    alpha = rng.uniform(0.05, 0.5, c).astype(np.float32)
    y = O.conv3d(f8(x), f8(w), f8(b), s_, p).astype(np.float32)
    d = dev()
    d.set_option("conv_split", split)
    # training-mode BatchNorm coefficients of y and the reduced sums of its backward, in float64
    yc = np.moveaxis(f8(y), 1, -1).reshape(-1, c)
    M = yc.shape[0]
    mean, var = yc.mean(0), yc.var(0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    scale, shift = f8(gamma) * invstd, f8(beta) - mean * f8(gamma) * invstd
    u = yc * scale + shift
    du = np.moveaxis(f8(dout), 1, -1).reshape(-1, c) * np.where(u > 0, 1.0, f8(alpha))
    xhat = (yc - mean) * invstd
    sums = np.concatenate([du.sum(0), (du * xhat).sum(0)])
    dy_ref = scale * (du - sums[:c] / M - xhat * sums[c:] / M)
    dy_ref = np.moveaxis(dy_ref.reshape(N, D, H, W, c), -1, 1)
    dx_ref = O.conv3d_dgrad(dy_ref, f8(w), x.shape, s_, p)
    dw_ref, _ = O.conv3d_wgrad(dy_ref, f8(x), k, s_, p)

    xt, yt, dot = t_from_ncdhw(x), t_from_ncdhw(y), t_from_ncdhw(dout)
    wp, bp = vec(w.ravel()), vec(b)
    cv = {n_: vec(v.astype(np.float32)) for n_, v in dict(scale=scale, shift=shift, alpha=alpha, mean=mean, invstd=invstd,
                                                            gamma=gamma, sums=sums).items()}
    desc = _desc(k, s_, p)
    nx = int(d.lib.msk_conv3d_xform_bytes(d.ctx, desc, xt.msk(), c))
    nb = int(d.lib.msk_conv3d_bwd_bnact_bytes(d.ctx, desc, xt.msk(), yt.msk()))
    assert (nb > 0) == (c >= 32) and (nx > 0) == (c >= 32)
    xf = d.malloc(nx) if nx else None
    ybuf = d.malloc(nb) if nb else None
    ytmp = t_empty(N, c, D, H, W, fill=0.0)
    d.call("msk_conv3d_fwd_ex", desc, xt.msk(), vp(wp), vp(bp), ytmp.msk(), None, vp(xf))   # fills xf for x
    # the reduce pass the backward always starts with: its maxima bound |dy| (needed by the fused forms under split 2)
    maxes, sums_dev = (vec(np.zeros(128, np.float32)), vec(np.zeros(3 * c, np.float32))) if c % 4 == 0 else (None, None)
    if maxes:
        from medicalseg_amd._lib import NULL_TENSOR
        d.call("msk_affine_act_bwd_reduce_ex", yt.msk(), vp(cv["scale"]), vp(cv["shift"]), NULL_TENSOR, vp(cv["alpha"]),
               vp(cv["mean"]), vp(cv["invstd"]), dot.msk(), vp(sums_dev), vp(maxes))
        mx = d.d2h(maxes, (2, 64), np.float32).max(axis=1)      # two amax arrays: the maximum of each counts
        assert abs(mx[0] - np.abs(du).max()) <= 1e-6 * np.abs(du).max() and abs(mx[1] - np.abs(xhat).max()) < 1e-4 * np.abs(xhat).max()
        got_sums = d.d2h(sums_dev, (2 * c,), np.float32)
        assert np.abs(got_sums - sums).max() < 1e-4 * np.abs(sums).max()
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    res = {}
    try:
        for form, opt in (("one", 1), ("split", 2), ("three", 0)):
            d.set_option("bwd_fuse", opt)
            dyt = t_empty(N, c, D, H, W, fill=9.0)
            dxt = t_empty(N, c, D, H, W, fill=3.0)
            dw = vec(np.full(w.size, 0.5, np.float32))
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_bwd_bnact", desc, xt.msk(), vp(wp), yt.msk(), vp(cv["scale"]), vp(cv["shift"]), vp(cv["alpha"]),
                   vp(cv["mean"]), vp(cv["invstd"]), vp(cv["gamma"]), dot.msk(), vp(cv["sums"]), C.c_double(float(M)),
                   dyt.msk(), dxt.msk(), 0, vp(dw), 0, vp(xf), vp(ybuf), vp(maxes))
            d.sync()
            d.prof_enable(False)
            rep = d.prof_report()
            res[form] = (t_to_ncdhw(dxt), d.d2h(dw, (w.size,), np.float32).reshape(w.shape), t_to_ncdhw(dyt), rep)
    finally:
        d.prof_enable(False)
        d.set_option("bwd_fuse", -1)
    eligible = c >= 32
    (dx3, dw3, dy3, rep3) = res["three"]
    assert "affine_act_bwd_apply" in rep3 and not any(t.startswith("wbf_tin_dual") or t.endswith("_bn_k") for t in rep3), rep3
    assert rel_err(dy3, dy_ref) < 2e-6
    tol = _conv_tol(c * K ** 3)
    tag_sfx = "_h2" if split == 2 and eligible else ""
    for form, tags in (("one", ("wbf_tin_dual_k",)), ("split", ("wbf_tin_bn_k", "wbf_ty_bn_k"))):
        dxf, dwf, dyf, repf = res[form]
        assert all((t in repf) == eligible for t in tags), (form, repf)
        assert ("affine_act_bwd_apply" in repf) == (not eligible), (form, repf)
        if eligible:
            assert np.all(dyf == 9.0)                       # dy never written in the fused forms
        print("%s: dx err %.2e (three %.2e) | dw err %.2e (three %.2e)" % (form, rel_err(dxf, dx_ref), rel_err(dx3, dx_ref),
                                                                         rel_err(dwf, dw_ref), rel_err(dw3, dw_ref)))
        assert rel_err(dxf, dx_ref) < tol and rel_err(dwf, dw_ref) < 2 * _conv_tol(M)
        assert rel_err(dxf, dx3) < 2e-6 and rel_err(dwf, dw3) < 2e-6     # the forms agree to fp32 rounding
        if eligible:
            assert ("wbf_gemm%s_k" % tag_sfx) in repf and ("wbf_wgrad%s_k" % tag_sfx) in repf, repf
    assert rel_err(dx3, dx_ref) < tol and rel_err(dw3, dw_ref) < 2 * _conv_tol(M)
    # accumulate semantics of both outputs
    dxt = t_from_ncdhw(dx_ref.astype(np.float32))
    dw = vec(dw_ref.astype(np.float32).ravel())
    dyt = t_empty(N, c, D, H, W, fill=9.0)
    d.call("msk_conv3d_bwd_bnact", desc, xt.msk(), vp(wp), yt.msk(), vp(cv["scale"]), vp(cv["shift"]), vp(cv["alpha"]),
           vp(cv["mean"]), vp(cv["invstd"]), vp(cv["gamma"]), dot.msk(), vp(cv["sums"]), C.c_double(float(M)),
           dyt.msk(), dxt.msk(), 1, vp(dw), 1, vp(xf), vp(ybuf), vp(maxes))
    d.sync()
    assert rel_err(t_to_ncdhw(dxt), 2 * dx_ref) < tol
    assert rel_err(d.d2h(dw, (w.size,), np.float32).reshape(w.shape), 2 * dw_ref) < 2 * _conv_tol(M)
    # round 5, msk_conv3d_bwd_bnact_split: the accumulated data gradient stored as two DENSE channel halves when the one-kernel
    # matrix stage runs it (forced here: wbf_fuse 2; c <= 64) -- bitwise the interleaved result of the same call without the split,
    # the interleaved buffer itself left at its old values; otherwise split_done = 0 and dx holds the result
    if eligible and c <= 64:
        d.set_option("wbf_fuse", 2)
        try:
            old = (0.25 * dx_ref).astype(np.float32)
            base = t_from_ncdhw(old)
            dw2 = vec(np.zeros(w.size, np.float32))
            d.call("msk_conv3d_bwd_bnact", desc, xt.msk(), vp(wp), yt.msk(), vp(cv["scale"]), vp(cv["shift"]), vp(cv["alpha"]),
                   vp(cv["mean"]), vp(cv["invstd"]), vp(cv["gamma"]), dot.msk(), vp(cv["sums"]), C.c_double(float(M)),
                   dyt.msk(), base.msk(), 1, vp(dw2), 0, vp(xf), vp(ybuf), vp(maxes))
            want = t_to_ncdhw(base)
            inter = t_from_ncdhw(old)
            lo, hi = t_empty(N, c // 2, D, H, W, fill=7.0), t_empty(N, c // 2, D, H, W, fill=7.0)
            done = C.c_int(-1)
            d.call("msk_conv3d_bwd_bnact_split", desc, xt.msk(), vp(wp), yt.msk(), vp(cv["scale"]), vp(cv["shift"]), vp(cv["alpha"]),
                   vp(cv["mean"]), vp(cv["invstd"]), vp(cv["gamma"]), dot.msk(), vp(cv["sums"]), C.c_double(float(M)),
                   dyt.msk(), inter.msk(), 1, vp(dw2), 0, vp(xf), vp(ybuf), vp(maxes), lo.msk(), hi.msk(), C.byref(done))
            d.sync()
            if c == 32 and K == 5:
                assert done.value == 1, done.value        # the one-kernel stage takes these (packed weights of all points within an XCD's L2)
            if done.value == 1:
                got = np.concatenate([t_to_ncdhw(lo), t_to_ncdhw(hi)], axis=1)
                assert np.array_equal(got, want)
                assert np.array_equal(t_to_ncdhw(inter), old)
            else:                                         # three-stage form (e.g. 64 channels x three bf16 pieces: 4.9 MB of packed weights)
                assert done.value == 0 and np.array_equal(t_to_ncdhw(inter), want)
                assert np.all(t_to_ncdhw(lo) == 7.0) and np.all(t_to_ncdhw(hi) == 7.0)
            # option dst_split 0: the plain behaviour through the same entry point
            d.set_option("dst_split", 0)
            inter2 = t_from_ncdhw(old)
            d.call("msk_conv3d_bwd_bnact_split", desc, xt.msk(), vp(wp), yt.msk(), vp(cv["scale"]), vp(cv["shift"]), vp(cv["alpha"]),
                   vp(cv["mean"]), vp(cv["invstd"]), vp(cv["gamma"]), dot.msk(), vp(cv["sums"]), C.c_double(float(M)),
                   dyt.msk(), inter2.msk(), 1, vp(dw2), 0, vp(xf), vp(ybuf), vp(maxes), lo.msk(), hi.msk(), C.byref(done))
            assert done.value == 0 and np.array_equal(t_to_ncdhw(inter2), want)
        finally:
            d.set_option("dst_split", 1)
            d.set_option("wbf_fuse", 1)


@pytest.mark.parametrize("case", [(32, 5, (2, 16, 32, 16), 1.0), (64, 5, (1, 8, 16, 16), 1e-6), (128, 5, (1, 8, 16, 8), 1e3),
                                  (32, 3, (2, 16, 16, 16), 1e-7)])
def test_wbf_fp16_two_piece_split_matches_oracle(case):
    """Option "conv_split" 2 (msk_wbf.h): operands as two fp16 pieces with a scaled residual, three MFMAs per fp32 product.
    Forward on O(1) activations; data and weight gradients on dy of the given magnitude (1e-7 .. 1e3) -- the pipelines scale dy
    by a power of two derived from its device-side maximum, so the result must be as accurate as for O(1) values, inside the
    tolerance of every other convolution kernel; the bf16x3 pipeline on the same inputs is printed next to it."""
    c, K, (N, D, H, W), mag = case
    k, s_, p = (K,) * 3, (1, 1, 1), (K // 2,) * 3
    d = dev()
    rng = np.random.default_rng(c + K)
    f8 = lambda a: a.astype(np.float64)
    x = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    x = np.where(x > 0, x, 0.25 * x).astype(np.float32)
    w = (rng.standard_normal((c, c) + k) * np.sqrt(2.0 / (c * K ** 3))).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    dy = (rng.standard_normal((N, c, D, H, W)) * mag).astype(np.float32)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    dw_ref, _ = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, dyt, wp, bp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel()), vec(b)
    desc = _desc(k, s_, p)
    M = N * D * H * W
    errs = {}
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    for split in (3, 2):
        d.set_option("conv_split", split)
        yt, dxt = t_empty(N, c, D, H, W, fill=7.0), t_empty(N, c, D, H, W, fill=3.0)
        dw = vec(np.zeros(w.size, np.float32))
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", desc, xt.msk(), vp(wp), vp(bp), yt.msk())
        d.call("msk_conv3d_dgrad", desc, dyt.msk(), vp(wp), dxt.msk(), 0)
        d.call("msk_conv3d_wgrad", desc, xt.msk(), dyt.msk(), vp(dw), None, 0)
        d.sync()
        d.prof_enable(False)
        rep = d.prof_report()
        sfx = "_h2" if split == 2 else ""
        assert rep.get("wbf_gemm%s_k" % sfx, (0, 0))[0] == 2 and rep.get("wbf_wgrad%s_k" % sfx, (0, 0))[0] == 1, rep
        errs[split] = (rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref),
                       rel_err(d.d2h(dw, (w.size,), np.float32).reshape(w.shape), dw_ref))
    print("c=%d K=%d |dy|~%g: fwd/dgrad/wgrad err  bf16x3 %.2e %.2e %.2e | fp16x2 %.2e %.2e %.2e" % ((c, K, mag) + errs[3] + errs[2]))
    for split in (3, 2):
        assert errs[split][0] < _conv_tol(c * K ** 3) and errs[split][1] < _conv_tol(c * K ** 3)
        assert errs[split][2] < 2 * _conv_tol(M)


@pytest.mark.parametrize("mag", [1e-30, 1e-12, 1e6, 1e15, 0.0])
def test_wbf_fp16_two_piece_split_scale_invariance(mag):
    """The fp16 two-piece pipelines scale every operand by a power of two taken from its device-side maximum, so a
    convolution of (mag * x, w / mag-ish weights) must be as accurate as at magnitude 1: activations, weights and gradients
    of 1e-30 ... 1e15 (far outside fp16's 6e-8 ... 65504), and an all-zero tensor (amax = 0 -> scale 1, exact zeros out)."""
    c, K, (N, D, H, W) = 32, 5, (1, 16, 16, 16)
    k, s_, p = (K,) * 3, (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", 2)
    rng = np.random.default_rng(5)
    f8 = lambda a: a.astype(np.float64)
    x = (rng.standard_normal((N, c, D, H, W)) * mag).astype(np.float32)
    wmag = 1e-3 if mag >= 1 else 1e3
    w = (rng.standard_normal((c, c) + k) * wmag).astype(np.float32)
    dy = (rng.standard_normal((N, c, D, H, W)) * mag).astype(np.float32)
    y_ref = O.conv3d(f8(x), f8(w), None, s_, p)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    dw_ref, _ = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, dyt, wp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel())
    desc = _desc(k, s_, p)
    yt, dxt = t_empty(N, c, D, H, W, fill=7.0), t_empty(N, c, D, H, W, fill=3.0)
    dw = vec(np.full(w.size, 9.0, np.float32))
    d.call("msk_conv3d_fwd", desc, xt.msk(), vp(wp), None, yt.msk())
    d.call("msk_conv3d_dgrad", desc, dyt.msk(), vp(wp), dxt.msk(), 0)
    d.call("msk_conv3d_wgrad", desc, xt.msk(), dyt.msk(), vp(dw), None, 0)
    d.sync()
    got = (t_to_ncdhw(yt), t_to_ncdhw(dxt), d.d2h(dw, (w.size,), np.float32).reshape(w.shape))
    if mag == 0.0:
        assert all(np.all(g == 0.0) for g in got)
        return
    assert all(np.all(np.isfinite(g)) for g in got)
    errs = [rel_err(got[0], y_ref), rel_err(got[1], dx_ref), rel_err(got[2], dw_ref)]
    print("mag %g: fwd / dgrad / wgrad err %.2e %.2e %.2e" % ((mag,) + tuple(errs)))
    assert errs[0] < _conv_tol(c * K ** 3) and errs[1] < _conv_tol(c * K ** 3) and errs[2] < 2 * _conv_tol(N * D * H * W)


@pytest.mark.parametrize("case", [(32, 20, (1, 12, 32, 24)), (20, 32, (1, 12, 32, 24)), (24, 40, (2, 8, 16, 16)), (48, 32, (1, 9, 16, 20)),
                                  (64, 20, (1, 9, 16, 20), 3), (20, 64, (2, 8, 16, 16), 3)])
def test_wbf_channel_padding_wrapper(case):
    """Channel counts that are not multiples of 32 (out_tr.conv1 of the 20-class MRI model, vnet.py:165: 32 -> 20 and the data
    gradient 20 -> 32) run through the fp16 two-piece pipeline on a zero-padded problem (gconv_wbf_padded, msk_conv.hip):
    forward with bias, data gradient plain and accumulating, against the float64 oracle; the padding is exact, so the
    tolerance is the pipeline's own."""
    cin, cout, (N, D, H, W) = case[:3]
    ks = case[3] if len(case) > 3 else 5          # 3: the 3x3x3 deep-supervision heads (vnet_deepsup.py:247-256)
    k, s_, p = (ks,) * 3, (1, 1, 1), (ks // 2,) * 3
    d = dev()
    d.set_option("conv_split", 2)
    d.set_option("wbf_pad_min_voxels", 0)
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * ks ** 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    y_ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s_, p)
    dx_ref = O.conv3d_dgrad(dy.astype(np.float64), w.astype(np.float64), x.shape, s_, p)
    xt, dyt, wp, bp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel()), vec(b)
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    try:
        yt = t_empty(N, cout, D, H, W, fill=9.0)
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.prof_enable(False)
        rep = d.prof_report()
        assert "wbf_gemm_h2_k" in rep and "pad_weights" in rep, rep
        # (round 4: a narrow SOURCE is read in place -- the transform takes its missing channels as zeros -- no padded copy)
        assert "pad_channels" not in rep and ("unpad_channels" in rep) == (cout % 32 != 0), rep
        e_f = rel_err(t_to_ncdhw(yt), y_ref)
        dxt = t_empty(N, cin, D, H, W, fill=5.0)
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        d.prof_enable(False)
        rep = d.prof_report()
        assert "wbf_gemm_h2_k" in rep, rep
        assert "pad_channels" not in rep and ("unpad_channels" in rep) == (cin % 32 != 0), rep
        e_d = rel_err(t_to_ncdhw(dxt), dx_ref)
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
        e_a = rel_err(t_to_ncdhw(dxt), 2 * dx_ref)
        # weight gradient through the padded pipeline as well (on the side stream, with its own scratch)
        dw_ref = O.conv3d_wgrad(dy.astype(np.float64), x.astype(np.float64), k, s_, p)[0]
        dwp, dbp = vec(np.full(w.size, 0.25, np.float32)), vec(np.zeros(cout))
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        d.prof_enable(False)
        rep = d.prof_report()
        assert "unpad_dw" in rep and "wbf_wgrad_h2_k" in rep, rep
        assert ("pad_channels" in rep) == (cin % 32 != 0), rep   # a narrow dy is read in place, a narrow x still padded
        from helpers import vec_back
        e_w = rel_err(vec_back(dwp, w.size).reshape(w.shape), dw_ref)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 1)
        e_wa = rel_err(vec_back(dwp, w.size).reshape(w.shape), 2 * dw_ref)
        M = N * D * H * W
        assert e_w < 2 * _conv_tol(M) and e_wa < 2 * _conv_tol(M), (e_w, e_wa)
        # below the size threshold the wrapper steps aside
        d.set_option("wbf_pad_min_voxels", 1 << 30)
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.prof_enable(False)
        assert "pad_weights" not in d.prof_report()
        assert rel_err(t_to_ncdhw(yt), y_ref) < _conv_tol(cin * 125)
    finally:
        d.prof_enable(False)
        d.set_option("wbf_pad_min_voxels", 1 << 18)
    print("padded pipeline: fwd %.2e dgrad %.2e acc %.2e" % (e_f, e_d, e_a))
    assert e_f < _conv_tol(cin * 125) and e_d < _conv_tol(cout * 125) and e_a < _conv_tol(cout * 125)


# ---------------------------------------------------------------------------------------------------------
# round 3: matrix stage + output transform in one kernel (wbf_gemm_fused_k), packed-weight cache
# ---------------------------------------------------------------------------------------------------------
FUSED_CASES = [
    # (Cin, Cout, (N, D, H, W))  -- option "wbf_fuse" 2 forces the one-kernel form below its size threshold
    (32, 32, (2, 16, 32, 16)),      # CN 32, 16 x 16 tiles, exact fit
    (32, 32, (1, 30, 60, 8)),       # ragged d and h tiles: masked rows in the epilogue and in the statistics
    (64, 64, (1, 20, 13, 16)),      # CN 64 (two column fragments per workgroup), transform along D
    (64, 32, (1, 15, 30, 12)),      # 4 chunks
    (32, 32, (1, 32, 32, 9)),       # ragged transform axis: stores and statistics of the one-kernel form stop at the volume's edge
    (64, 64, (2, 16, 16, 6)),
]


@pytest.mark.parametrize("split", [2, 3])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_wbf_fused_output_stage_matches_oracle_and_three_stage_form(case, split):
    """Forward (bias, BatchNorm statistics of y), data gradient (plain and accumulating) and the PReLU epilogue of the
    one-kernel form against the float64 oracle, and against the three-stage form (wbf_tout_k) on the same inputs."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", split)
    rng = np.random.default_rng(cin * 7 + cout + H)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    alpha = rng.uniform(0.05, 0.5, cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    wp, bp, ap = vec(w.ravel()), vec(b), vec(alpha)
    cd = _desc(k, s_, p)
    res = {}
    try:
        for mode in (2, 0):
            d.set_option("wbf_fuse", mode)
            yt, ya = t_empty(N, cout, D, H, W, fill=7.0), t_empty(N, cout, D, H, W, fill=7.0)
            dxt = t_empty(N, cin, D, H, W, fill=3.0)
            stats = vec(np.zeros(2 * cout, np.float32))
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_fwd_ex", cd, xt.msk(), vp(wp), vp(bp), yt.msk(), vp(stats), None)
            d.prof_enable(False)
            rep = d.prof_report()
            # the one-kernel form has no output-transform launch; it needs the packed weights of all 8 points in an XCD's L2
            # (<= 3.5 MB: 64 channels with the three-piece split are over it and keep the three stages)
            one_kernel = mode == 2 and 8 * 25 * (cin // 16) * 2 * split * cout * 16 <= (3584 << 10)
            assert ("wbf_tout_k" in rep) == (not one_kernel), rep
            d.call("msk_conv3d_fwd_act", cd, xt.msk(), vp(wp), vp(bp), vp(ap), ya.msk())
            d.call("msk_conv3d_dgrad", cd, dyt.msk(), vp(wp), dxt.msk(), 0)
            dx1 = t_to_ncdhw(dxt)
            d.call("msk_conv3d_dgrad", cd, dyt.msk(), vp(wp), dxt.msk(), 1)
            res[mode] = (t_to_ncdhw(yt), vec_back(stats, 2 * cout), t_to_ncdhw(ya), dx1, t_to_ncdhw(dxt))
    finally:
        d.set_option("wbf_fuse", 1)
    tol_f, tol_d = _conv_tol(cin * 125), _conv_tol(cout * 125)
    act_ref = np.where(y_ref > 0, y_ref, alpha.reshape(1, -1, 1, 1, 1) * y_ref)
    M = N * D * H * W
    for mode, (y, st, ya, dx1, dx2) in res.items():
        assert rel_err(y, y_ref) < tol_f, (mode, rel_err(y, y_ref))
        assert rel_err(ya, act_ref) < tol_f
        assert rel_err(dx1, dx_ref) < tol_d and rel_err(dx2, 2 * dx_ref) < tol_d
        # statistics record: mean[C], M2[C] of the STORED values
        yc = np.moveaxis(f8(y), 1, 0).reshape(cout, -1)
        np.testing.assert_allclose(st[:cout], yc.mean(1), rtol=0, atol=2e-6 * np.abs(yc).max())
        np.testing.assert_allclose(st[cout:], yc.var(1) * M, rtol=2e-5)
    # the two forms differ in the summation order of the output transform (and in split-K on the small cases)
    assert rel_err(res[2][0], res[0][0]) < 8e-6 and rel_err(res[2][3], res[0][3]) < 8e-6   # two fp32-class results (each 1-2.5e-6 from the oracle)
    print(f"\nfused {case} split {split}: fwd {rel_err(res[2][0], y_ref):.2e} (three-stage {rel_err(res[0][0], y_ref):.2e})  "
          f"dgrad {rel_err(res[2][3], dx_ref):.2e} ({rel_err(res[0][3], dx_ref):.2e})")


def test_wbf_packed_weight_cache_follows_every_weight_write():
    """The packed weights of a layer are cached per weight tensor; every entry point that can change the tensor must
    invalidate them: h2d, d2d, memset, the optimizer kernels (which also rebuild them in one launch), free + reuse."""
    cin = cout = 32
    N, D, H, W = 1, 16, 16, 8
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", 2)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    xt, yt = t_from_ncdhw(x), t_empty(N, cout, D, H, W)
    cd = _desc(k, s_, p)
    f8 = lambda a: a.astype(np.float64)
    mkw = lambda sc: (rng.standard_normal((cout, cin) + k) * sc / np.sqrt(cin * 125)).astype(np.float32)
    b = np.zeros(cout, np.float32)
    bp = vec(b)
    count = cout * cin * 125

    def check(wp, w, what):
        d.call("msk_conv3d_fwd", cd, xt.msk(), vp(wp), vp(bp), yt.msk())
        e = rel_err(t_to_ncdhw(yt), O.conv3d(f8(x), f8(w), f8(b), s_, p))
        assert e < _conv_tol(cin * 125), (what, e)

    w1 = mkw(1.0)
    wp = vec(w1.ravel())
    check(wp, w1, "first use")
    check(wp, w1, "cached")
    w2 = mkw(37.0)                      # a different maximum: a stale amax would show as well
    d.h2d(wp, w2.ravel())
    check(wp, w2, "after h2d")
    w3 = mkw(1e-3)
    src = vec(w3.ravel())
    d.d2d(wp, src, count * 4)
    check(wp, w3, "after d2d")
    d.memset(wp, 0, count * 4)
    check(wp, np.zeros_like(w1), "after memset")
    # optimizer kernel: w <- w - lr * (g + wd * w), velocity 0, momentum 0
    d.h2d(wp, w1.ravel())
    g = mkw(1.0)
    gp, vel = vec(g.ravel()), vec(np.zeros(count, np.float32))
    check(wp, w1, "before sgd")
    d.call("msk_sgd_momentum", vp(wp), vp(gp), vp(vel), C.c_size_t(count), C.c_float(0.5), C.c_float(0.0), C.c_float(0.0),
           C.c_float(1.0))
    d.prof_reset()
    d.prof_enable(True)
    check(wp, w1 - 0.5 * g, "after sgd")
    d.prof_enable(False)
    assert "wbf_pack_weights" not in d.prof_report(), d.prof_report()   # rebuilt by the optimizer kernel's epilogue, not here
    m1, m2 = vec(np.zeros(count, np.float32)), vec(np.zeros(count, np.float32))
    d.call("msk_adam", vp(wp), vp(gp), vp(m1), vp(m2), C.c_size_t(count), C.c_float(1e-2), C.c_float(0.9), C.c_float(0.999),
           C.c_float(1e-8), C.c_double(0.9), C.c_double(0.999), C.c_float(0.0), C.c_float(1.0))
    w_adam = d.d2h(wp, (cout, cin) + k, np.float32)
    assert np.abs(w_adam - (w1 - 0.5 * g)).max() > 1e-3
    check(wp, w_adam, "after adam")
    # free + a new allocation (very likely at the same address) with other weights
    d.free(wp)
    w5 = mkw(5.0)
    wp2 = vec(w5.ravel())
    check(wp2, w5, "after free + malloc")
    # cache off: same results
    d.set_option("wbf_pack_cache", 0)
    try:
        check(wp2, w5, "cache off")
    finally:
        d.set_option("wbf_pack_cache", 1)


def test_wbf_packed_weight_cache_free_of_an_arena_and_foreign_writes():
    """Round-3 advisor: (a) a weight tensor that lives INSIDE a larger allocation (the parameter arena) -- msk_free of the arena
    must drop its cache row, not just invalidate it: the next optimizer call rebuilds every stale row in use and would read
    freed memory (and the row's packed buffer would leak).  (b) weights changed behind the library's back (here: a library
    pass that is not on the invalidation list, msk_copy_scale) need msk_weights_changed -- the stale result is demonstrated,
    then the call fixes it."""
    cin = cout = 32
    N, D, H, W = 1, 16, 16, 8
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", 2)
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    xt, yt = t_from_ncdhw(x), t_empty(N, cout, D, H, W)
    cd = _desc(k, s_, p)
    f8 = lambda a: a.astype(np.float64)
    mkw = lambda sc: (rng.standard_normal((cout, cin) + k) * sc / np.sqrt(cin * 125)).astype(np.float32)
    bp = vec(np.zeros(cout, np.float32))
    count = cout * cin * 125

    def run(wp):
        d.call("msk_conv3d_fwd", cd, xt.msk(), vp(wp), vp(bp), yt.msk())
        return t_to_ncdhw(yt)

    def err(wp, w):
        return rel_err(run(wp), O.conv3d(f8(x), f8(w), np.zeros(cout), s_, p))

    # (a) weights at an offset inside an "arena"
    off = 4096
    arena = d.malloc(off * 4 + count * 4 + 1024)
    w1 = mkw(1.0)
    d.h2d(arena + off * 4, w1.ravel())
    assert err(arena + off * 4, w1) < _conv_tol(cin * 125)
    d.free(arena)
    # an optimizer call on OTHER memory: rebuilds every stale live row in use -- the freed row must be gone by now
    other = vec(mkw(1.0).ravel())
    g, vel = vec(np.zeros(count, np.float32)), vec(np.zeros(count, np.float32))
    d.call("msk_sgd_momentum", vp(other), vp(g), vp(vel), C.c_size_t(count), C.c_float(0.1), C.c_float(0.0), C.c_float(0.0), C.c_float(1.0))
    d.sync()
    arena2 = d.malloc(off * 4 + count * 4 + 1024)        # very likely the same address range
    w2 = mkw(9.0)
    d.h2d(arena2 + off * 4, w2.ravel())
    assert err(arena2 + off * 4, w2) < _conv_tol(cin * 125)
    # (b) a write the library does not track: copy_scale of another tensor over the weights
    from medicalseg_amd.device import Tensor
    w3 = mkw(0.25)
    src = vec(w3.ravel())
    wp = arena2 + off * 4
    as_t = lambda ptr: Tensor(d, ptr, 1, 1, 1, count // 4, 4, 4, None)
    d.call("msk_copy_scale", as_t(src).msk(), None, as_t(wp).msk(), 0)
    assert np.array_equal(d.d2h(wp, (count,), np.float32), w3.ravel())
    stale = err(wp, w3)
    assert stale > 1e-2, stale                              # the convolution still used the packed copy of w2: the documented hazard
    d.call("msk_weights_changed", vp(wp), C.c_size_t(count * 4))
    assert err(wp, w3) < _conv_tol(cin * 125)
    d.free(arena2)


# ---------------------------------------------------------------------------------------------------------
# round 3: the fp16 two-piece operand format under ADVERSARIAL intra-tensor dynamic range
# ---------------------------------------------------------------------------------------------------------
def _bulk_err(got, ref, region):
    """max |got - ref| over `region` relative to max |ref| over the same region"""
    g, r = got[region].astype(np.float64), ref[region]
    return float(np.abs(g - r).max() / (np.abs(r).max() + 1e-300))


def _three_kernel_sets(d, fn):
    """fn() -> result, under the product format (conv_split 2), the exact bf16 x 3 split and the exact-fp32 Winograd kernels"""
    out = {}
    try:
        for name, split, bf3 in (("fp16x2", 2, 1), ("bf16x3", 3, 1), ("fp32", 2, 0)):
            d.set_option("conv_split", split)
            d.set_option("wino_bf3", bf3)
            out[name] = fn()
    finally:
        d.set_option("conv_split", 2)
        d.set_option("wino_bf3", 1)
    return out


@pytest.mark.parametrize("log2_range", [10, 14, 17, 20])
def test_wbf_fp16_split_sparse_outliers_keep_the_bulk(log2_range):
    """x (and dy) = a unit-scale bulk plus a slab of outliers 2^r above it.  The tensor-wide power-of-two scale is set by
    the outliers; the claim under test is that the BULK -- outputs whose receptive field holds no outlier -- is still
    computed at fp32 class: error of the bulk relative to the bulk's own maximum, against the float64 oracle, next to the
    exact bf16 x 3 split and the exact-fp32 Winograd kernels on the same data."""
    c, K, (N, D, H, W) = 32, 5, (1, 24, 16, 16)
    k, s_, p = (K,) * 3, (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(100 + log2_range)
    f8 = lambda a: a.astype(np.float64)
    big = float(2.0 ** log2_range)
    x = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    x = np.where(x > 0, x, 0.25 * x).astype(np.float32)
    dy = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    # outliers: 1 % of the voxels of the planes d < 4 (all channels of a hit voxel)
    hit = rng.random((N, 1, 4, H, W)) < 0.01
    x[:, :, :4] = np.where(hit, x[:, :, :4] * big, x[:, :, :4])
    dy[:, :, :4] = np.where(hit, dy[:, :, :4] * big, dy[:, :, :4])
    w = (rng.standard_normal((c, c) + k) * np.sqrt(2.0 / (c * K ** 3))).astype(np.float32)
    y_ref = O.conv3d(f8(x), f8(w), None, s_, p)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    bulk = (slice(None), slice(None), slice(8, None))        # d >= 8: no outlier within the 5^3 support
    xt, dyt, wp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel())
    desc = _desc(k, s_, p)

    def run():
        yt, dxt = t_empty(N, c, D, H, W, fill=7.0), t_empty(N, c, D, H, W, fill=3.0)
        d.call("msk_conv3d_fwd", desc, xt.msk(), vp(wp), None, yt.msk())
        d.call("msk_conv3d_dgrad", desc, dyt.msk(), vp(wp), dxt.msk(), 0)
        y, dx = t_to_ncdhw(yt), t_to_ncdhw(dxt)
        return (_bulk_err(y, y_ref, bulk), _bulk_err(dx, dx_ref, bulk), rel_err(y, y_ref), rel_err(dx, dx_ref))

    e = _three_kernel_sets(d, run)
    print("\noutliers 2^%d above the bulk: BULK error fwd / dgrad   fp16x2 %.2e %.2e | bf16x3 %.2e %.2e | fp32 Winograd %.2e %.2e"
          "   (whole tensor: %.2e %.2e | %.2e %.2e | %.2e %.2e)" % (
              (log2_range,) + e["fp16x2"][:2] + e["bf16x3"][:2] + e["fp32"][:2] + e["fp16x2"][2:] + e["bf16x3"][2:] + e["fp32"][2:]))
    tol = _conv_tol(c * K ** 3)
    # whole tensor (dominated by the outliers): the usual bound
    assert e["fp16x2"][2] < tol and e["fp16x2"][3] < tol
    # the bulk: fp32 class -- the tolerance of every convolution kernel, and no worse than 4x the exact-fp32 kernels' bulk error
    for i in (0, 1):
        assert e["fp16x2"][i] < tol, (i, e)
        assert e["fp16x2"][i] < 4 * max(e["fp32"][i], 5e-7), (i, e)


def test_wbf_fp16_split_real_loss_gradient_with_sparse_classes():
    """dy shaped like what backward really feeds the LUConv layers: the CE + Dice logit gradient of a label map with a class
    of 0.05 % of the voxels (large gradients there, ~1e-6 elsewhere), spread over 32 channels, plus an x with a dead channel
    (exact zeros) and a near-constant one.  Data gradient and weight gradient against the float64 oracle; the quiet region
    (no rare-class voxel in the support) is reported separately."""
    c, K, (N, D, H, W) = 32, 5, (1, 16, 32, 32)
    k, s_, p = (K,) * 3, (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(7)
    f8 = lambda a: a.astype(np.float64)
    logits = rng.standard_normal((N, 3, D, H, W)) * 2.0
    labels = (rng.random((N, D, H, W)) < 0.3).astype(np.int32)           # classes 0 / 1 everywhere ...
    rare = np.zeros((N, D, H, W), bool)
    rare[:, :3, :6, :6] = rng.random((N, 3, 6, 6)) < 0.08                 # ... class 2 on a few voxels of one corner
    labels[rare] = 2
    logits[:, 2][~rare] -= 12.0                                           # confidently not class 2 elsewhere: tiny gradients
    _, _, dz = O.MixedLossOracle()(logits, labels)
    mix = rng.standard_normal((3, c)) / np.sqrt(3.0)
    dy = np.einsum("nkdhw,kc->ncdhw", dz, mix).astype(np.float32)
    x = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    x = np.where(x > 0, x, 0.25 * x).astype(np.float32)
    x[:, 3] = 0.0                                                         # a dead channel
    x[:, 5] = 1e-3 + 1e-7 * x[:, 5]                                       # a nearly constant one
    w = (rng.standard_normal((c, c) + k) * np.sqrt(2.0 / (c * K ** 3))).astype(np.float32)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    dw_ref, _ = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    quiet = (slice(None), slice(None), slice(8, None))                    # d >= 8: the rare class is >= 5 planes away
    print("\n|dy|: max %.3e, median %.3e, quiet-region max %.3e" % (np.abs(dy).max(), np.median(np.abs(dy)), np.abs(dy[quiet]).max()))
    xt, dyt, wp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel())
    desc = _desc(k, s_, p)

    def run():
        dxt = t_empty(N, c, D, H, W, fill=3.0)
        dw = vec(np.zeros(w.size, np.float32))
        d.call("msk_conv3d_dgrad", desc, dyt.msk(), vp(wp), dxt.msk(), 0)
        d.call("msk_conv3d_wgrad", desc, xt.msk(), dyt.msk(), vp(dw), None, 0)
        d.sync()
        dx = t_to_ncdhw(dxt)
        dwv = d.d2h(dw, (w.size,), np.float32).reshape(w.shape)
        return (rel_err(dx, dx_ref), _bulk_err(dx, dx_ref, quiet), rel_err(dwv, dw_ref),
                float(np.abs(dwv[:, 3]).max()))                           # the dead channel's weight gradient: exactly 0

    e = _three_kernel_sets(d, run)
    for name, v in e.items():
        print("%-7s dgrad %.2e (quiet region %.2e)  wgrad %.2e  dead-channel |dw| %.1e" % ((name,) + v))
    tol = _conv_tol(c * K ** 3)
    v = e["fp16x2"]
    assert v[0] < tol and v[2] < 2 * _conv_tol(N * D * H * W) and v[3] == 0.0
    assert v[1] < tol and v[1] < 4 * max(e["fp32"][1], 5e-7), e


@pytest.mark.parametrize("log2_range", [10, 13, 17, 20])
def test_wbf_fp16_split_quiet_channels_in_the_weight_gradient(log2_range):
    """Per-CHANNEL dynamic range: half of the channels of x and of dy are 2^r below the other half (a nearly dead feature
    next to a loud one).  The weight gradient sums over all positions, so there is no quiet REGION -- but dw[co, ci] of a
    quiet (co, ci) pair is built from quiet operands only; its error is taken relative to the quiet block's own maximum."""
    c, K, (N, D, H, W) = 32, 5, (1, 16, 16, 16)
    k, s_, p = (K,) * 3, (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(200 + log2_range)
    f8 = lambda a: a.astype(np.float64)
    small = float(2.0 ** -log2_range)
    x = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, c, D, H, W)).astype(np.float32)
    x[:, 16:] *= small
    dy[:, 16:] *= small
    dw_ref, _ = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    desc = _desc(k, s_, p)
    blocks = {"loud x loud": (slice(0, 16), slice(0, 16)), "quiet x loud": (slice(16, 32), slice(0, 16)),
              "quiet x quiet": (slice(16, 32), slice(16, 32))}

    def run():
        dw = vec(np.zeros(dw_ref.size, np.float32))
        d.call("msk_conv3d_wgrad", desc, xt.msk(), dyt.msk(), vp(dw), None, 0)
        d.sync()
        g = d.d2h(dw, (dw_ref.size,), np.float32).reshape(dw_ref.shape)
        return tuple(_bulk_err(g, dw_ref, b) for b in blocks.values())

    e = _three_kernel_sets(d, run)
    print("\nchannels 2^-%d: weight-gradient error per block (%s)   fp16x2 %.2e %.2e %.2e | bf16x3 %.2e %.2e %.2e | fp32 Winograd "
          "%.2e %.2e %.2e" % ((log2_range, ", ".join(blocks)) + e["fp16x2"] + e["bf16x3"] + e["fp32"]))
    # Round 4: the weight-gradient kernel renormalises every dy channel by its own maximum (measured by the kernel that writes
    # the A dy transform) and applies both in-register factors 2^-11 of the cross terms on that side: fp32 class at EVERY
    # channel range (round 3: exact to 2^13 between channels, then one bit lost per factor of two -- 3.8e-5 / 3.1e-4 of the
    # quiet block's maximum at 2^17 / 2^20).  Option wgrad_renorm 0 = the per-tensor scales alone, for the A/B printed here.
    tol = 2 * _conv_tol(N * D * H * W)
    for i in range(3):
        assert e["fp16x2"][i] < tol, (i, e)
        assert e["fp16x2"][i] < 4 * max(e["fp32"][i], 5e-7), (i, e)
    d.set_option("conv_split", 2)
    d.set_option("wgrad_renorm", 0)
    try:
        off = run()
    finally:
        d.set_option("wgrad_renorm", 1)
    print("   per-tensor scales only (wgrad_renorm 0): %.2e %.2e %.2e" % off)
    if log2_range >= 17:
        assert max(off) > 4 * max(e["fp16x2"])      # the renormalisation is what makes the difference


@pytest.mark.parametrize("case", [(32, 32, (2, 16, 32, 16)), (1, 16, (1, 12, 16, 32)), (16, 32, (1, 8, 8, 8))])
def test_fwd_ex3_finalisation_in_the_merge_is_bitwise_the_two_call_form(case):
    """msk_conv3d_fwd_ex3(fin) == msk_conv3d_fwd_ex2 + msk_bn_finalize(world 1): same statistics record, and bit-identical
    mean / invstd / scale / shift / running statistics (the claim of include/msegk.h); the third case takes its statistics
    from msk_bn_stats_fin (no epilogue statistics for that kernel)."""
    import ctypes as C
    from medicalseg_amd._lib import MskBnFin
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    d.set_option("conv_split", 2)
    rng = np.random.default_rng(cin + 3 * cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 2.0, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    xt, wp, bp, gp, bep = t_from_ncdhw(x), vec(w.ravel()), vec(b), vec(gamma), vec(beta)
    cd = _desc(k, s_, p)
    M = float(N * D * H * W)
    res = []
    for fused in (0, 1):
        yt = t_empty(N, cout, D, H, W, fill=7.0)
        bufs = {n_: vec(np.full(cout, 3.0 if n_ in ("rm", "rv") else 9.0, np.float32)) for n_ in ("rm", "rv", "mean", "invstd", "scale", "shift")}
        stats = vec(np.zeros(2 * cout, np.float32))
        if fused:
            fin = MskBnFin(gp, bep, 1e-5, 0.9, M, bufs["rm"], bufs["rv"], bufs["mean"], bufs["invstd"], bufs["scale"], bufs["shift"])
            d.call("msk_conv3d_fwd_ex3", cd, xt.msk(), vp(wp), vp(bp), yt.msk(), vp(stats), None, None, C.byref(fin))
        else:
            d.call("msk_conv3d_fwd_ex2", cd, xt.msk(), vp(wp), vp(bp), yt.msk(), vp(stats), None, None)
            d.call("msk_bn_finalize", vp(stats), 1, C.c_double(M), cout, vp(gp), vp(bep), C.c_float(1e-5), C.c_float(0.9),
                   vp(bufs["rm"]), vp(bufs["rv"]), vp(bufs["mean"]), vp(bufs["invstd"]), vp(bufs["scale"]), vp(bufs["shift"]))
        res.append({n_: vec_back(p_, cout) for n_, p_ in bufs.items()} | {"stats": vec_back(stats, 2 * cout), "y": t_to_ncdhw(yt)})
    for n_ in res[0]:
        assert np.array_equal(res[0][n_], res[1][n_]), n_
    yc = np.moveaxis(res[1]["y"].astype(np.float64), 1, 0).reshape(cout, -1)
    np.testing.assert_allclose(res[1]["mean"], yc.mean(1), rtol=0, atol=3e-6 * np.abs(yc).max())
    np.testing.assert_allclose(res[1]["rv"], 0.9 * 3.0 + 0.1 * yc.var(1), rtol=2e-5)


def test_bwd_reduce_pg_is_bitwise_reduce_plus_param_grads():
    """msk_affine_act_bwd_reduce_pg == msk_affine_act_bwd_reduce_ex + msk_affine_act_param_grads (accumulating), maxes from the
    zeroed ring arrays instead of a memset."""
    import ctypes as C
    from medicalseg_amd._lib import NULL_TENSOR
    c, (N, D, H, W) = 32, (2, 8, 16, 16)
    d = dev()
    rng = np.random.default_rng(9)
    mk = lambda: t_from_ncdhw(rng.standard_normal((N, c, D, H, W)).astype(np.float32))
    x, dout = mk(), mk()
    v = lambda lo=0.5, hi=1.5: vec(rng.uniform(lo, hi, c).astype(np.float32))
    scale, shift, alpha, mean, invstd = v(), v(-0.5, 0.5), v(0.1, 0.4), v(-0.2, 0.2), v()
    outs = []
    for pg in (0, 1):
        sums = vec(np.zeros(4 * c, np.float32))
        grads = [vec(np.full(c, 2.0, np.float32)) for _ in range(3)]        # dgamma, dbeta, dalpha start at 2: accumulate
        if pg:
            maxes = d.amax_new(2)
            d.call("msk_affine_act_bwd_reduce_pg", x.msk(), vp(scale), vp(shift), NULL_TENSOR, vp(alpha), vp(mean), vp(invstd),
                   dout.msk(), vp(sums), vp(maxes), 0, vp(grads[0]), vp(grads[1]), vp(grads[2]))
        else:
            maxes = vec(np.full(128, 5.0, np.float32))                       # stale contents: the _ex form clears them
            d.call("msk_affine_act_bwd_reduce_ex", x.msk(), vp(scale), vp(shift), NULL_TENSOR, vp(alpha), vp(mean), vp(invstd),
                   dout.msk(), vp(sums), vp(maxes))
            d.call("msk_affine_act_param_grads", c, vp(sums), vp(grads[0]), vp(grads[1]), vp(grads[2]), 1)
        mx = vec_back(maxes, 128)
        outs.append([vec_back(sums, 3 * c)] + [vec_back(g, c) for g in grads] + [np.array([mx[:64].max(), mx[64:].max()])])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[1][4][0] > 0 and np.abs(outs[1][1] - 2.0).max() > 1e-3


@pytest.mark.parametrize("split", [2, 1])   # fp16 two-piece / single fp16 (conv_fp16): both scale dy by its maximum
def test_gradient_maximum_from_the_pass_that_writes_dy(split):
    """msk_affine_act_bwd_apply_amax writes the same dy as msk_affine_act_bwd_apply and folds max |dy| into an amax array; the
    gradient entry points given that array (msk_conv3d_dgrad_ex / msk_conv3d_wgrad_ex2) return bit-identical results to the
    plain calls, which measure dy with a pass of their own."""
    import ctypes as C
    from medicalseg_amd._lib import NULL_TENSOR
    c, (N, D, H, W) = 32, (2, 8, 16, 16)
    d = dev()
    d.set_option("conv_split", 2)
    d.set_option("conv_fp16", 1 if split == 1 else 0)
    d.set_option("wgrad_async", 0)
    try:
        rng = np.random.default_rng(17 + split)
        mk = lambda s=1.0: t_from_ncdhw((s * rng.standard_normal((N, c, D, H, W))).astype(np.float32))
        y, dout, x = mk(), mk(3e-4), mk()
        v = lambda lo, hi: vec(rng.uniform(lo, hi, c).astype(np.float32))
        scale, shift, alpha, mean, invstd, gamma = v(0.5, 1.5), v(-0.5, 0.5), v(0.1, 0.4), v(-0.2, 0.2), v(0.5, 1.5), v(0.5, 1.5)
        sums = vec((rng.standard_normal(3 * c) * 10).astype(np.float32))
        M = float(N * D * H * W)
        outs = []
        for with_amax in (0, 1):
            dy = t_empty(N, c, D, H, W, fill=0.0)
            am = d.amax_new(1) if with_amax else None
            args = (y.msk(), vp(scale), vp(shift), NULL_TENSOR, vp(alpha), vp(mean), vp(invstd), vp(gamma), dout.msk(), vp(sums),
                    C.c_double(M), 1, dy.msk(), NULL_TENSOR, 0)
            if with_amax:
                d.call("msk_affine_act_bwd_apply_amax", *args, vp(am))
            else:
                d.call("msk_affine_act_bwd_apply", *args)
            dyh = t_to_ncdhw(dy)
            if with_amax:
                assert vec_back(am, 64).max() == np.abs(dyh).max()
            w = (rng.standard_normal((c, c, 5, 5, 5)) / 60).astype(np.float32) if not outs else w
            wp = vec(w.ravel()) if not outs else wp
            dx = t_empty(N, c, D, H, W, fill=1.0)
            dw, db = vec(np.zeros(w.size, np.float32)), vec(np.zeros(c, np.float32))
            cd = _desc((5,) * 3, (1,) * 3, (2,) * 3)
            if with_amax:
                d.call("msk_conv3d_dgrad_ex", cd, dy.msk(), vp(wp), dx.msk(), 1, vp(am))
                d.call("msk_conv3d_wgrad_ex2", cd, x.msk(), dy.msk(), vp(dw), vp(db), 0, None, vp(am))
            else:
                d.call("msk_conv3d_dgrad", cd, dy.msk(), vp(wp), dx.msk(), 1)
                d.call("msk_conv3d_wgrad", cd, x.msk(), dy.msk(), vp(dw), vp(db), 0)
            outs.append((dyh, t_to_ncdhw(dx), vec_back(dw, w.size), vec_back(db, c)))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        assert np.abs(outs[0][2]).max() > 0
    finally:
        d.set_option("conv_fp16", 0)
        d.set_option("wgrad_async", 1)


@pytest.mark.parametrize("mode", ["plain", "alpha", "alpha+input residual"])
def test_bwd_bnact_one_input_channel_evaluates_dy_in_the_weight_gradient(mode):
    """in_tr.conv1 (1 -> 16 channels, no data gradient): the weight-gradient kernel evaluates dy from (y, dout) itself -- same dw
    as msk_affine_act_bwd_apply + msk_conv3d_wgrad, one kernel less.  Through msk_conv3d_bwd_bnact (no residual; dy_scratch left
    alone) and through msk_conv3d_bwd_bnact_c1 with the unit's tiled one-channel input as the residual (vnet.py:75-78)."""
    import ctypes as C
    from medicalseg_amd._lib import NULL_TENSOR
    cout, (N, D, H, W) = 16, (2, 9, 18, 61)    # (a volume narrower than 3/4 of its 32-voxel W tiles is declined: the MRI slab)
    d = dev()
    rng = np.random.default_rng(23 + len(mode))
    x = rng.standard_normal((N, 1, D, H, W)).astype(np.float32)
    y = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    dout = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    xt, yt, dt = t_from_ncdhw(x), t_from_ncdhw(y), t_from_ncdhw(dout, ld=32)      # dout: a slice of a wider gradient buffer
    v = lambda lo, hi: vec(rng.uniform(lo, hi, cout).astype(np.float32))
    scale, shift, alpha, mean, invstd, gamma = v(0.5, 1.5), v(-0.5, 0.5), v(0.1, 0.4), v(-0.2, 0.2), v(0.5, 1.5), v(0.5, 1.5)
    sums = vec((rng.standard_normal(3 * cout) * 30).astype(np.float32))
    w = vec((rng.standard_normal(cout * 125) / 11).astype(np.float32))
    M = float(N * D * H * W)
    cd = _desc((5,) * 3, (1,) * 3, (2,) * 3)
    al = vp(alpha) if mode != "plain" else None
    res = xt.msk() if "residual" in mode else NULL_TENSOR
    # reference: the two separate calls
    dy = t_empty(N, cout, D, H, W, fill=0.0)
    d.call("msk_affine_act_bwd_apply", yt.msk(), vp(scale), vp(shift), res, al, vp(mean), vp(invstd), vp(gamma), dt.msk(),
           vp(sums), C.c_double(M), 1, dy.msk(), NULL_TENSOR, 0)
    dw_ref = vec(np.full(cout * 125, 0.5, np.float32))
    d.call("msk_conv3d_wgrad", cd, xt.msk(), dy.msk(), vp(dw_ref), None, 1)
    # the one call
    scratch = t_empty(N, cout, D, H, W, fill=-7.0)
    dw = vec(np.full(cout * 125, 0.5, np.float32))
    d.prof_reset()
    d.prof_enable(True)
    if "residual" in mode:
        rc = d.lib.msk_conv3d_bwd_bnact_c1(d.ctx, cd, xt.msk(), yt.msk(), vp(scale), vp(shift), al, vp(mean), vp(invstd), res, dt.msk(),
                                           vp(sums), C.c_double(M), vp(dw), 1)
        assert rc == 0
        # a residual that is not the input itself is declined, nothing launched
        other = t_from_ncdhw(x.copy())
        assert d.lib.msk_conv3d_bwd_bnact_c1(d.ctx, cd, xt.msk(), yt.msk(), vp(scale), vp(shift), al, vp(mean), vp(invstd), other.msk(),
                                             dt.msk(), vp(sums), C.c_double(M), vp(dw), 1) == 1
    else:
        d.call("msk_conv3d_bwd_bnact", cd, xt.msk(), vp(w), yt.msk(), vp(scale), vp(shift), al, vp(mean), vp(invstd), vp(gamma),
               dt.msk(), vp(sums), C.c_double(M), scratch.msk(), NULL_TENSOR, 0, vp(dw), 1, None, None, None)
    d.sync()
    d.prof_enable(False)
    tags = d.prof_report()
    assert tags["wgrad_c1_mfma"][0] == 1 and "affine_act_bwd_apply" not in tags, tags
    a, b = vec_back(dw, cout * 125), vec_back(dw_ref, cout * 125)
    assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    assert np.all(t_to_ncdhw(scratch) == -7.0)
    dy64 = t_to_ncdhw(dy).astype(np.float64)
    dw_or, _ = O.conv3d_wgrad(dy64, x.astype(np.float64), (5,) * 3, (1,) * 3, (2,) * 3)
    assert rel_err(a.reshape(dw_or.shape) - 0.5, dw_or) < _conv_tol(M) * 2


@pytest.mark.parametrize("split", [2, 3])
@pytest.mark.parametrize("case", [(32, 32, (1, 16, 16, 16)), (64, 128, (1, 8, 16, 8)), (128, 64, (1, 8, 8, 7)), (256, 256, (1, 8, 8, 4))])
def test_wbf_pack_weights_lds_form_is_bitwise_the_elementwise_form(case, split):
    """Round 5: the packed weight image built through the LDS tile (whole-run loads, 16-byte stores: wbf_pack_weights_lds_k) is
    the image of the one-thread-per-element kernel bit for bit -- forward AND data gradient (swap / flip of the canonical
    tensor, a permuted transform axis in the third case) give identical outputs under option wbf_pack_lds 1 / 0."""
    cin, cout, (N, D, H, W) = case
    d = dev()
    d.set_option("conv_split", split)
    try:
        rng = np.random.default_rng(cin + cout)
        k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
        x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
        dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
        xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
        wp, bp = vec(w.ravel()), vec(np.zeros(cout, np.float32))
        outs = []
        for lds in (1, 0):
            d.set_option("wbf_pack_lds", lds)
            d.h2d(wp, w.ravel())                     # invalidates the cached image: the next call packs with the selected kernel
            yt, dxt = t_empty(N, cout, D, H, W), t_empty(N, cin, D, H, W)
            d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
            d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
            outs.append((t_to_ncdhw(yt), t_to_ncdhw(dxt)))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        f8 = lambda a: a.astype(np.float64)
        assert rel_err(outs[0][0], O.conv3d(f8(x), f8(w), None, s_, p)) < _conv_tol(cin * 125)
    finally:
        d.set_option("wbf_pack_lds", 1)
        d.set_option("conv_split", 2)


@pytest.mark.parametrize("c,size", [(32, 64), (64, 64)])
def test_pipelined_one_kernel_matrix_stage_is_bitwise_reproducible(c, size):
    """Round 6: wbf_gemm_fused_k's pipelined form issues its LDS-DMA fills by inline assembly and synchronises with hand-counted
    s_waitcnt vmcnt(N) + raw s_barrier (msk_conv_wbf.hip, BL == 2) -- a miscounted wait or a slot reused too early would show as a
    result that depends on timing.  The same forward + accumulating data gradient, six times on unchanged inputs, must be bitwise
    identical, and the one-kernel form must be the one that ran (32 channels: four phases per stage; 64: five)."""
    import hashlib
    d = dev()
    d.set_option("conv_split", 2)
    rng = np.random.default_rng(3)
    n = 2
    vox = n * size ** 3
    from medicalseg_amd.device import Tensor
    mk = lambda: Tensor(d, d.malloc(vox * c * 4), n, size, size, size, c, c, None)
    x, y, dx = mk(), mk(), mk()
    d.h2d(x.ptr, rng.standard_normal(vox * c, dtype=np.float32))
    w = d.malloc(c * c * 125 * 4)
    d.h2d(w, (rng.standard_normal(c * c * 125) * 0.01).astype(np.float32))
    b = d.small(c)
    cd = _desc((5, 5, 5), (1, 1, 1), (2, 2, 2))
    d.prof_reset()
    d.set_option("prof_only_halo", 0)
    d.set_option("prof_shapes", 1)
    d.prof_enable(True)
    hashes = set()
    try:
        for _ in range(6):
            d.call("msk_conv3d_fwd", cd, x.msk(), C.c_void_p(w), C.c_void_p(b), y.msk())
            d.memset(dx.ptr, 0, vox * c * 4)
            d.call("msk_conv3d_dgrad", cd, y.msk(), C.c_void_p(w), dx.msk(), 1)
            hashes.add(hashlib.sha256(d.d2h(y.ptr, (vox * c,), np.float32).tobytes() + d.d2h(dx.ptr, (vox * c,), np.float32).tobytes()).hexdigest())
    finally:
        d.prof_enable(False)
        d.set_option("prof_shapes", 0)
        for t in (x, y, dx):
            d.free(t.ptr)
        d.free(w)
    tags = d.prof_report()
    assert any(k.startswith("wbf_gemm_h2_k") and "fused" in k for k in tags), sorted(tags)
    assert len(hashes) == 1, "results differ between runs: %d distinct" % len(hashes)
