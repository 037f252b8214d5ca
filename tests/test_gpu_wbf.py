"""Three-stage Winograd F(4,5) x bf16x3 pipeline (msk_conv_wbf.hip: wbf_tin_k -> wbf_gemm_k -> wbf_tout_k) against
the float64 oracle, through the C ABI.

The matrix stage multiplies fp32 operands that were split EXACTLY into three bf16 pieces with six bf16 products per
fp32 product (fp32 accumulate); the claim under test is fp32-class accuracy: the same tolerance as every other
convolution kernel of this repo (tests/test_gpu_ops.py: 8e-6 * sqrt(K/1000 + 1) of max|ref|), and in addition an
error no larger than 1.5x that of the exact-fp32 Winograd kernels it replaces on the same inputs."""
import numpy as np
import pytest

from helpers import dev, rel_err, t_empty, t_from_ncdhw, t_to_ncdhw, vec, vp

pytestmark = pytest.mark.gpu

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
]


@pytest.mark.parametrize("case", WBF_CASES)
def test_wbf_fwd_dgrad_match_oracle(case):
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
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
        assert rep.get("wbf_gemm_k", (0, 0))[0] >= 1, rep          # the pipeline really ran (the forward at least;
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
    print(f"\nwbf {case}: fwd {e_f:.2e} (fp32 kernels {o_f:.2e})  dgrad {e_d:.2e} ({o_d:.2e})  acc {e_acc:.2e}")
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
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wbf_wgrad_matches_oracle(case):
    """wbf_wgrad_k (transposing LDS reads, v_mfma_f32_16x16x32_bf16, six products per fp32 product) + its split-K / G^T
    reduce against the float64 oracle, fresh and accumulating, and next to the exact-fp32 kernels on the same inputs."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
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
        assert rep.get("wbf_wgrad_k", (0, 0))[0] == 2, rep
        d.set_option("wino_bf3", 0)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        old = d.d2h(dwp, (nw,), np.float32).reshape(dw_ref.shape)
    finally:
        d.set_option("wino_bf3", 1)
        d.set_option("wgrad_async", 1)
    e, e2, eo = rel_err(got, dw_ref), rel_err(got2, 2 * dw_ref), rel_err(old, dw_ref)
    M = N * D * H * W
    print(f"\nwbf wgrad {case}: {e:.2e} (accumulating {e2:.2e}; fp32 kernels {eo:.2e})")
    assert e < _conv_tol(M) and e2 < _conv_tol(M)
    assert e < 4e-6


@pytest.mark.parametrize("case", [(32, 32, (2, 16, 32, 16)), (64, 128, (1, 8, 16, 8)), (32, 64, (1, 14, 30, 8)),
                                  (16, 16, (1, 8, 8, 8))])
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


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("case", K3_CASES)
def test_wbf_3x3x3_pipeline(case, fp16):
    """The same three-stage pipeline for 3 x 3 x 3 convolutions (Winograd F(4,3), 6 points; UNet3D's DoubleConvs):
    forward, data gradient and weight gradient against the float64 oracle,
      fp16 = 0: exact bf16x3 operands -> the fp32-class tolerance of every other convolution kernel;
      fp16 = 1: option "conv_fp16" (the fp16 matrix path of BASELINE configs[3]): fp16 operands in the Winograd domain,
                fp32 accumulate.  STATED fp16 TOLERANCE: 3e-3 of max|ref| (operand rounding 2^-11 ~ 4.9e-4 per value,
                amplified ~3x by the output transform; measured 1.0e-3 .. 1.8e-3)."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    d = dev()
    rng = np.random.default_rng(cin * 3 + cout + D + fp16)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
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
    print(f"\nwbf 3^3 {case} fp16={fp16}: fwd {e_f:.2e} dgrad {e_d:.2e} wgrad {e_w:.2e}")
    if fp16:
        assert e_f < 3e-3 and e_d < 3e-3 and e_w < 3e-3
    else:
        assert e_f < _conv_tol(cin * 27) and e_d < _conv_tol(cout * 27) and e_w < _conv_tol(N * D * H * W) * 2
