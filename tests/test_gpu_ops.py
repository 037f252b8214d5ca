"""Per-op parity: HIP kernels (through the C ABI) vs the numpy oracle, on the GPU.

Tolerances (fp32 kernels vs float64 oracle, stated per the north-star's "fp32 tolerance"):
  convolutions: max abs err <= 8e-6 * max|ref| * sqrt(K/1000 + 1)   (K = reduction length; about 3x the largest
  error any kernel shows: direct fp32 MFMA ~5e-7, bf16x3 Winograd F(4,5) <= 2.7e-6, fp32 Winograd F(4,5) <= 4.5e-6)
  elementwise / BN / loss: 1e-5 relative to max|ref|.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import dev, rel_err, t_empty, t_from_ncdhw, t_to_ncdhw, vec, vec_back, vp

pytestmark = pytest.mark.gpu

from oracle import vnet_numpy as O  # noqa: E402


def _desc(k, s, p):
    from medicalseg_amd._lib import MskConvDesc
    return MskConvDesc(*k, *s, *p)


CONV_CASES = [
    # (Cin, Cout, k, s, p, (N, D, H, W))
    (32, 32, (5, 5, 5), (1, 1, 1), (2, 2, 2), (2, 6, 9, 35)),     # halo MFMA, TW=32, ragged tiles
    (64, 64, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 5, 10, 16)),    # TW=16
    (128, 96, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 4, 8, 8)),     # TW=8, Cout not multiple of 64
    (1, 16, (5, 5, 5), (1, 1, 1), (2, 2, 2), (2, 8, 8, 12)),      # in_tr: Cin=1
    (1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 6, 9, 34)),      # UNet3D first convolution: conv_c1_h2 (two channel tiles), wgrad_c1 in two blocks of 16
    (1, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 5, 8, 16)),
    (32, 3, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 6, 8, 33)),      # out_tr: Cout=3
    (32, 4, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 5, 9, 17)),      # two-voxel VALU kernel: CN even, ragged tiles
    (32, 1, (5, 5, 5), (1, 1, 1), (2, 2, 2), (2, 4, 8, 16)),
    (32, 2, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 9, 7, 40)),
    (20, 20, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 5, 6)),      # out_tr.conv2
    (3, 3, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 5, 6)),
    (4, 2, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 19, 33, 47)),     # wgrad_pw_small_k: several blocks, CA != CB
    (20, 20, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 16, 64, 64)),   # MRI out_tr.conv2 (20 classes): streaming pointwise_mid kernel
    (32, 3, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 9, 33, 17)),     # UNet3D head 32 -> ncls (and ncls -> 32 as its data gradient): pointwise_thin
    (16, 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 8, 8, 40)),
    (24, 12, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 5, 6, 7)),      # pointwise_mid with unequal sides
    (16, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 8, 8, 8)),      # down conv
    (16, 32, (2, 2, 4), (2, 2, 1), (0, 0, 0), (1, 8, 8, 12)),     # MRI anisotropic down conv
    (16, 32, (2, 2, 4), (2, 2, 1), (0, 0, 0), (1, 32, 64, 12)),   # the same at a size the fine-level weight-gradient kernel takes (M >= 4096)
    (32, 64, (2, 2, 2), (2, 2, 1), (0, 0, 0), (2, 16, 32, 9)),    # MRI level 2: kernel (2, 2, 2), stride (2, 2, 1)
    (8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 6, 7, 9)),        # 3x3x3 (deep-sup head shape class)
    (32, 40, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 7, 30, 4)),     # MRI slab depth 4 -> halo tile <4,16,4>
    (24, 32, (5, 5, 5), (1, 1, 1), (2, 2, 2), (2, 15, 31, 2)),    # MRI slab depth 2 -> halo tile <8,16,2>
    (64, 64, (5, 5, 5), (1, 1, 1), (2, 2, 2), (1, 8, 16, 9)),     # MRI level 2 (W = 9)
    (16, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 7, 30, 4)),     # 3^3 on the narrow tiles
    (64, 3, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 6, 9, 34)),      # VNetDeepSup out_tr64 (lung, ncls 3)
    (128, 20, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 5, 9, 12)),    # VNetDeepSup out_tr128 (MRI, ncls 20)
    (256, 5, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 4, 6, 6)),      # VNetDeepSup out_tr256
    (5, 7, (3, 2, 1), (2, 1, 1), (1, 0, 0), (1, 7, 6, 5)),        # odd everything -> reference kernels
    (8, 16, (2, 2, 4), (2, 2, 4), (0, 0, 0), (1, 5, 6, 13)),      # k == s with 16 taps: two tap groups in wgrad_ks, floor dims
    (4, 8, (3, 3, 3), (3, 3, 3), (0, 0, 0), (2, 7, 9, 10)),       # k == s with 27 taps (4 tap groups, last one partial)
    (48, 40, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 9, 6, 34)),     # k == s, two channel tiles each side, odd D
    (16, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 4, 6, 64)),     # round 5: gconv_ks_lds_k (<= 16 source channels, runs of 32 destination voxels along W)
    (16, 64, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 4, 4, 128)),    # ... two N tiles per workgroup, two runs per row
    (8, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 2, 6, 64)),      # ... one 8-channel chunk
]


def _conv_tol(K):
    return 8e-6 * np.sqrt(K / 1000.0 + 1.0)


@pytest.mark.parametrize("impl", [0, 1, 7])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(case, impl):
    cin, cout, k, s, p, (N, D, H, W) = case
    d = dev()
    d.set_option("conv_impl", impl)
    try:
        rng = np.random.default_rng(hash((cin, cout, k, s)) % 2**31)
        x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * np.prod(k))).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y_ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, p)
        od, oh, ow = y_ref.shape[2:]
        xt, yt = t_from_ncdhw(x), t_empty(N, cout, od, oh, ow, fill=7.0)
        wp, bp = vec(w.ravel()), vec(b)
        d.call("msk_conv3d_fwd", _desc(k, s, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        K = cin * int(np.prod(k))
        assert rel_err(t_to_ncdhw(yt), y_ref) < _conv_tol(K)

        dy = rng.standard_normal(y_ref.shape).astype(np.float32)
        dx_ref = O.conv3d_dgrad(dy.astype(np.float64), w.astype(np.float64), x.shape, s, p)
        dw_ref, db_ref = O.conv3d_wgrad(dy.astype(np.float64), x.astype(np.float64), k, s, p)
        dyt = t_from_ncdhw(dy)
        dxt = t_empty(N, cin, D, H, W, fill=3.0)
        d.call("msk_conv3d_dgrad", _desc(k, s, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        assert rel_err(t_to_ncdhw(dxt), dx_ref) < _conv_tol(cout * int(np.prod(k)))
        # accumulate semantics
        d.call("msk_conv3d_dgrad", _desc(k, s, p), dyt.msk(), vp(wp), dxt.msk(), 1)
        assert rel_err(t_to_ncdhw(dxt), 2 * dx_ref) < _conv_tol(cout * int(np.prod(k)))

        dwp, dbp = vec(np.full(w.size, 0.5, np.float32)), vec(np.full(cout, 0.25, np.float32))
        d.call("msk_conv3d_wgrad", _desc(k, s, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        M = N * od * oh * ow
        assert rel_err(vec_back(dwp, w.size).reshape(w.shape), dw_ref) < _conv_tol(M) * 2
        assert rel_err(vec_back(dbp, cout), db_ref) < 1e-5 * np.sqrt(M / 1000 + 1)
        d.call("msk_conv3d_wgrad", _desc(k, s, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 1)
        assert rel_err(vec_back(dwp, w.size).reshape(w.shape), 2 * dw_ref) < _conv_tol(M) * 2
    finally:
        d.set_option("conv_impl", 0)


def test_mri_anisotropic_levels_take_the_streaming_kernels():
    """Round 4: kernel (2, 2, 4) / stride (2, 2, 1) and (2, 2, 2) / (2, 2, 1) (vnet_mri_spine_seg_512_512_12_15k.yml:9-10) are not
    kernel == stride; they ran on the general parity-class / tap-row kernels (6.5 ms of the 33 ms MRI step).  The streaming
    kernels never depended on kernel == stride for unpadded windows -- gconv_ks_fwd (forward, up-conv data gradient),
    wgrad_ks2 (both weight gradients) -- and gconv_kst is the transposed form with the overlap along W; the numbers are
    checked by the CONV_CASES / CONVT_CASES entries above, here that the product dispatch really selects them."""
    d = dev()
    rng = np.random.default_rng(3)
    k, s, p0 = (2, 2, 4), (2, 2, 1), (0, 0, 0)
    N, D, H, W = 1, 32, 64, 12
    x = rng.standard_normal((N, 16, D, H, W)).astype(np.float32)
    w = rng.standard_normal((32, 16) + k).astype(np.float32)
    xt, yt = t_from_ncdhw(x), t_empty(N, 32, D // 2, H // 2, W - 3)
    dyt = t_from_ncdhw(rng.standard_normal((N, 32, D // 2, H // 2, W - 3)).astype(np.float32))
    dxt = t_empty(N, 16, D, H, W)
    wp, dwp, dbp = vec(w.ravel()), vec(np.zeros(w.size, np.float32)), vec(np.zeros(32, np.float32))
    wt = rng.standard_normal((32, 16) + k).astype(np.float32)      # Conv3DTranspose weight [Cin = 32][Cout = 16]
    wtp, dwtp, dbtp = vec(wt.ravel()), vec(np.zeros(wt.size, np.float32)), vec(np.zeros(16, np.float32))
    d.set_option("prof_only_halo", 0)
    d.set_option("prof_shapes", 0)
    d.prof_reset()
    d.prof_enable(True)
    d.call("msk_conv3d_fwd", _desc(k, s, p0), xt.msk(), vp(wp), None, yt.msk())
    d.call("msk_conv3d_dgrad", _desc(k, s, p0), dyt.msk(), vp(wp), dxt.msk(), 0)
    d.call("msk_conv3d_wgrad", _desc(k, s, p0), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
    d.call("msk_convT3d_fwd", _desc(k, s, p0), dyt.msk(), vp(wtp), None, dxt.msk())
    d.call("msk_convT3d_dgrad", _desc(k, s, p0), xt.msk(), vp(wtp), yt.msk(), 0)
    d.call("msk_convT3d_wgrad", _desc(k, s, p0), dyt.msk(), xt.msk(), vp(dwtp), vp(dbtp), 0)
    d.sync()
    d.prof_enable(False)
    rep = d.prof_report()
    assert rep.get("gconv_ks_fwd", (0, 0))[0] == 2, rep          # conv forward + convT data gradient
    assert rep.get("gconv_kst", (0, 0))[0] == 2, rep             # conv data gradient + convT forward
    assert rep.get("wgrad_ks2_mfma", (0, 0))[0] == 2, rep
    assert "gconv_gather_mfma" not in rep and "wgrad_mfma" not in rep, rep


CONVT_CASES = [
    (32, 16, (2, 2, 2), (2, 2, 2), (2, 4, 4, 4)),
    (64, 16, (2, 2, 4), (2, 2, 1), (1, 4, 4, 9)),     # MRI up conv (overlap-add along W)
    (64, 16, (2, 2, 4), (2, 2, 1), (1, 16, 32, 9)),   # the same at a size the fine-level weight-gradient kernel takes
    (128, 32, (2, 2, 2), (2, 2, 1), (2, 8, 16, 8)),   # MRI level 2 up conv
    (6, 5, (3, 2, 2), (2, 2, 1), (1, 3, 4, 5)),
    # kernel == stride: taps-folded scatter kernel (msk_conv_scatter.hip)
    (64, 16, (2, 2, 2), (2, 2, 2), (1, 4, 8, 32)),    # up_tr32.up_conv class
    (128, 32, (2, 2, 2), (2, 2, 2), (1, 3, 5, 7)),    # two N groups, ragged M, two K passes
    (12, 5, (2, 2, 2), (2, 2, 2), (2, 3, 3, 5)),      # CK % 8 != 0, taps*CN % 32 != 0
    (16, 8, (2, 2, 1), (2, 2, 1), (1, 4, 4, 6)),      # anisotropic k == s
    (40, 24, (1, 2, 2), (1, 2, 2), (2, 5, 4, 3)),
    (16, 8, (2, 2, 4), (2, 2, 4), (1, 3, 3, 3)),      # 16 taps
]


@pytest.mark.parametrize("impl", [0, 1, 7, 22])   # 7: taps-folded scatter kernels at any size; 22: 7 with their fragment-shaped form
@pytest.mark.parametrize("case", CONVT_CASES)
def test_convT3d_fwd_dgrad_wgrad(case, impl):
    cin, cout, k, s, (N, D, H, W) = case
    d = dev()
    d.set_option("conv_impl", 7 if impl == 22 else impl)
    d.set_option("ks_legacy", 2 if impl == 22 else 0)
    try:
        rng = np.random.default_rng(cin * 131 + cout)
        x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
        w = (rng.standard_normal((cin, cout) + k) / np.sqrt(cin * np.prod(k))).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        x64, w64 = x.astype(np.float64), w.astype(np.float64)
        y_ref = O.conv_transpose3d(x64, w64, b.astype(np.float64), s)
        od, oh, ow = y_ref.shape[2:]
        xt, yt = t_from_ncdhw(x), t_empty(N, cout, od, oh, ow, fill=-1.0)
        wp, bp = vec(w.ravel()), vec(b)
        d.call("msk_convT3d_fwd", _desc(k, s, (0, 0, 0)), xt.msk(), vp(wp), vp(bp), yt.msk())
        assert rel_err(t_to_ncdhw(yt), y_ref) < 3e-5
        dy = rng.standard_normal(y_ref.shape).astype(np.float32)
        dx_ref = O.conv_transpose3d_dgrad(dy.astype(np.float64), w64, s)
        dw_ref, db_ref = O.conv_transpose3d_wgrad(dy.astype(np.float64), x64, k, s)
        dyt, dxt = t_from_ncdhw(dy), t_empty(N, cin, D, H, W, fill=9.0)
        d.call("msk_convT3d_dgrad", _desc(k, s, (0, 0, 0)), dyt.msk(), vp(wp), dxt.msk(), 0)
        assert rel_err(t_to_ncdhw(dxt), dx_ref) < 3e-5
        dwp, dbp = vec(np.zeros(w.size, np.float32)), vec(np.zeros(cout, np.float32))
        d.call("msk_convT3d_wgrad", _desc(k, s, (0, 0, 0)), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        assert rel_err(vec_back(dwp, w.size).reshape(w.shape), dw_ref) < 5e-5
        assert rel_err(vec_back(dbp, cout), db_ref) < 2e-5
    finally:
        d.set_option("conv_impl", 0)
        d.set_option("ks_legacy", 0)


KS2_CASES = [
    # (transposed, fine channels, coarse channels, k (= s), (N, coarse D, H, W), fine ld, coarse ld, bias folded into the kernel)
    (0, 16, 32, (2, 2, 2), (2, 16, 16, 17), 16, 32, True),     # down conv 16 -> 32: one row group, bias = column sums of dy
    (1, 16, 64, (2, 2, 2), (2, 16, 16, 17), 32, 64, True),     # up conv 64 -> 16 into a concat slice (ld 32): bias = sums of dy over the taps
    (0, 32, 64, (2, 2, 2), (1, 16, 16, 16), 32, 64, True),     # two row groups share dy; the first one carries the bias
    (1, 32, 128, (2, 2, 2), (1, 16, 16, 17), 64, 128, False),  # four row groups: a transposed conv's bias takes the separate pass
    (0, 12, 40, (2, 2, 2), (1, 15, 17, 19), 12, 48, True),     # 96 rows (a dead row tile), ragged column tiles, odd dims
    (1, 8, 20, (2, 2, 1), (2, 16, 16, 9), 8, 20, True),        # four taps, one 32-row tile, anisotropic
    (0, 16, 32, (2, 2, 2), (1, 4, 16, 16), 16, 32, True),      # M = 1024 < 4096: the one-tap-per-tile kernel + channel sums
]


@pytest.mark.parametrize("case", KS2_CASES)
def test_wgrad_ks2_fine_levels(case):
    """Kernel == stride weight gradient with (tap, channel) rows (msk_wgrad_ks.hip, wgrad_ks2_k) and the bias gradient folded
    into the same pass: against the float64 oracle, accumulate semantics, channel-slice strides, and which kernels ran."""
    tr, cf, cc, k, (N, D, H, W), ldf, ldc, folded = case
    d = dev()
    rng = np.random.default_rng(cf * 1000 + cc + tr)
    fD, fH, fW = D * k[0], H * k[1], W * k[2]
    fine = rng.standard_normal((N, cf, fD, fH, fW)).astype(np.float32)
    coarse = rng.standard_normal((N, cc, D, H, W)).astype(np.float32)
    ft, ct = t_from_ncdhw(fine, ld=ldf), t_from_ncdhw(coarse, ld=ldc)
    cd = _desc(k, k, (0, 0, 0))
    taps = int(np.prod(k))
    if tr:   # up conv: x = coarse (cc channels), dy = fine (cf channels); w [cc][cf][k]
        dw_ref, db_ref = O.conv_transpose3d_wgrad(fine.astype(np.float64), coarse.astype(np.float64), k, k)
        nw, nb, fn, args = cc * cf * taps, cf, "msk_convT3d_wgrad", (ct.msk(), ft.msk())
    else:    # down conv: x = fine, dy = coarse; w [cc][cf][k]
        dw_ref, db_ref = O.conv3d_wgrad(coarse.astype(np.float64), fine.astype(np.float64), k, k, (0, 0, 0))
        nw, nb, fn, args = cc * cf * taps, cc, "msk_conv3d_wgrad", (ft.msk(), ct.msk())
    M = N * D * H * W
    d.set_option("wgrad_async", 0)
    try:
        dwp, dbp = vec(np.full(nw, 0.5, np.float32)), vec(np.full(nb, 0.25, np.float32))
        d.prof_reset()
        d.prof_enable(True)
        d.call(fn, cd, *args, vp(dwp), vp(dbp), 0)
        d.sync()
        d.prof_enable(False)
        tags = set(d.prof_report())
        big = M >= 4096
        assert ("wgrad_ks2_mfma" in tags) == big and ("wgrad_ks_mfma" in tags) == (not big), tags
        assert ("channel_sum_partial" in tags) == (not (big and folded)), tags
        tol = _conv_tol(M) * 2
        assert rel_err(vec_back(dwp, nw).reshape(dw_ref.shape), dw_ref) < tol
        assert rel_err(vec_back(dbp, nb), db_ref) < 1e-5 * np.sqrt(M * (taps if tr else 1) / 1000 + 1)
        d.call(fn, cd, *args, vp(dwp), vp(dbp), 1)
        assert rel_err(vec_back(dwp, nw).reshape(dw_ref.shape), 2 * dw_ref) < tol
        assert rel_err(vec_back(dbp, nb), 2 * db_ref) < 1e-5 * np.sqrt(M * (taps if tr else 1) / 1000 + 1)
        # the kernel it replaces gives the same sums up to fp32 summation order
        d.set_option("ks_legacy", 1)
        dw2, db2 = vec(np.zeros(nw, np.float32)), vec(np.zeros(nb, np.float32))
        d.call(fn, cd, *args, vp(dw2), vp(db2), 0)
        assert rel_err(vec_back(dw2, nw).reshape(dw_ref.shape), dw_ref) < tol
    finally:
        d.set_option("ks_legacy", 0)
        d.set_option("wgrad_async", 1)


@pytest.mark.parametrize("impl", [0, 22])
def test_convT_scatter_into_concat_slice(impl):
    """Up-convolution forward into the first 16 channels of a 32-channel concat buffer (vnet.py:133-150: the zero-copy skip
    connection) at a size the LDS-staged scatter kernel takes by default, and the data gradient of a down-convolution
    accumulating into a tensor that already holds the skip gradient."""
    d = dev()
    d.set_option("ks_legacy", 2 if impl == 22 else 0)
    try:
        rng = np.random.default_rng(31)
        N, D, H, W = 1, 16, 32, 33
        x = rng.standard_normal((N, 64, D, H, W)).astype(np.float32)
        w = (rng.standard_normal((64, 16, 2, 2, 2)) / 16).astype(np.float32)
        b = rng.standard_normal(16).astype(np.float32)
        y_ref = O.conv_transpose3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), (2, 2, 2))
        xt, yt = t_from_ncdhw(x), t_empty(N, 16, 2 * D, 2 * H, 2 * W, ld=32, fill=-3.0)
        d.call("msk_convT3d_fwd", _desc((2,) * 3, (2,) * 3, (0,) * 3), xt.msk(), vp(vec(w.ravel())), vp(vec(b)), yt.msk())
        assert rel_err(t_to_ncdhw(yt), y_ref) < 3e-5
        raw = vec_back(yt.ptr, N * 8 * D * H * W * 32).reshape(-1, 32)
        assert np.all(raw[:, 16:] == -3.0)      # the skip half of the buffer is untouched
        # down conv 16 -> 32 data gradient, accumulating
        dy = rng.standard_normal((N, 32, D, H, W)).astype(np.float32)
        wd = (rng.standard_normal((32, 16, 2, 2, 2)) / 11).astype(np.float32)
        g0 = rng.standard_normal((N, 16, 2 * D, 2 * H, 2 * W)).astype(np.float32)
        dx_ref = O.conv3d_dgrad(dy.astype(np.float64), wd.astype(np.float64), g0.shape, (2,) * 3, (0,) * 3) + g0
        dxt = t_from_ncdhw(g0)
        d.call("msk_conv3d_dgrad", _desc((2,) * 3, (2,) * 3, (0,) * 3), t_from_ncdhw(dy).msk(), vp(vec(wd.ravel())), dxt.msk(), 1)
        assert rel_err(t_to_ncdhw(dxt), dx_ref) < _conv_tol(32 * 8)
    finally:
        d.set_option("ks_legacy", 0)


def test_conv_strided_channel_slice():
    """Tensors that are channel slices of wider buffers (zero-copy concat) work for convs."""
    d = dev()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 32, 4, 6, 33)).astype(np.float32)
    w = (rng.standard_normal((32, 32, 5, 5, 5)) / 60).astype(np.float32)
    y_ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), None, 1, 2)
    xt = t_from_ncdhw(x, ld=48)
    yt = t_empty(1, 32, 4, 6, 33, ld=40, fill=0.0)
    d.call("msk_conv3d_fwd", _desc((5,) * 3, (1,) * 3, (2,) * 3), xt.msk(), vp(vec(w.ravel())), None, yt.msk())
    assert rel_err(yt.numpy(), y_ref) < 5e-5


@pytest.mark.parametrize("C_,shape", [(16, (2, 9, 10, 11)), (3, (1, 7, 8, 9)), (256, (2, 4, 4, 4)), (20, (1, 5, 6, 7)),
                                      # dense 1/2/3/6-channel tensors: the float4-triple kernels of round 5 (voxels * C % 12 == 0) ...
                                      (3, (2, 8, 12, 20)), (2, (1, 6, 8, 9)), (6, (1, 4, 5, 6)), (1, (2, 6, 6, 8)),
                                      (3, (1, 7, 7, 9))])                      # ... and a size they decline (the scalar kernels)
def test_bn_stats_and_affine_act(C_, shape):
    d = dev()
    N, D, H, W = shape
    rng = np.random.default_rng(C_)
    x = (rng.standard_normal((N, C_, D, H, W)) * 3 + 50.0).astype(np.float32)  # large mean: cancellation test
    gamma, beta = rng.uniform(0.5, 1.5, C_).astype(np.float32), rng.standard_normal(C_).astype(np.float32)
    alpha = rng.uniform(0.1, 0.4, C_).astype(np.float32)
    res = rng.standard_normal(x.shape).astype(np.float32)
    rm, rv = rng.standard_normal(C_).astype(np.float32), rng.uniform(0.5, 2, C_).astype(np.float32)
    x64 = x.astype(np.float64)
    y_ref, xhat, mean, var, invstd = O.bn_train(x64, gamma.astype(np.float64), beta.astype(np.float64))
    out_ref = O.prelu(y_ref + res, alpha.astype(np.float64))

    xt, rt, ot = t_from_ncdhw(x), t_from_ncdhw(res), t_empty(N, C_, D, H, W)
    stats = vec(np.zeros(2 * C_))
    d.call("msk_bn_stats", xt.msk(), vp(stats))
    st = vec_back(stats, 2 * C_)
    M = N * D * H * W
    assert np.abs(st[:C_] - mean).max() < 1e-5 * 50
    assert rel_err(st[C_:] / M, var) < 2e-5
    g, b_, rmp, rvp = vec(gamma), vec(beta), vec(rm), vec(rv)
    sm, si, sc, sh = vec(np.zeros(C_)), vec(np.zeros(C_)), vec(np.zeros(C_)), vec(np.zeros(C_))
    d.call("msk_bn_finalize", vp(stats), 1, C.c_double(M), C_, vp(g), vp(b_), C.c_float(1e-5), C.c_float(0.9),
           vp(rmp), vp(rvp), vp(sm), vp(si), vp(sc), vp(sh))
    assert rel_err(vec_back(si, C_), invstd) < 2e-5
    assert rel_err(vec_back(rmp, C_), 0.9 * rm + 0.1 * mean) < 1e-5
    assert rel_err(vec_back(rvp, C_), 0.9 * rv + 0.1 * var) < 2e-5
    al = vec(alpha)
    d.call("msk_affine_act_fwd", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), ot.msk())
    assert np.abs(t_to_ncdhw(ot) - out_ref).max() < 2e-4  # values O(10) after (x-50)/3 scaling

    # backward
    dout = rng.standard_normal(x.shape).astype(np.float32)
    u = y_ref + res
    du, dalpha = O.prelu_bwd(dout.astype(np.float64), u, alpha.astype(np.float64))
    dx_ref, dg_ref, db_ref = O.bn_train_bwd(du, xhat, gamma.astype(np.float64), invstd)
    dt = t_from_ncdhw(dout)
    sums = vec(np.zeros(3 * C_))
    d.call("msk_affine_act_bwd_reduce", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), vp(sm), vp(si), dt.msk(), vp(sums))
    s_ = vec_back(sums, 3 * C_)
    assert rel_err(s_[:C_], db_ref) < 1e-4
    assert rel_err(s_[C_:2 * C_], dg_ref) < 1e-4
    assert rel_err(s_[2 * C_:], dalpha) < 1e-4
    dxt, drt = t_empty(N, C_, D, H, W), t_empty(N, C_, D, H, W, fill=1.0)
    d.call("msk_affine_act_bwd_apply", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), vp(sm), vp(si), vp(g), dt.msk(),
           vp(sums), C.c_double(M), 1, dxt.msk(), drt.msk(), 1)
    assert rel_err(t_to_ncdhw(dxt), dx_ref) < 2e-4
    assert rel_err(t_to_ncdhw(drt), du + 1.0) < 1e-5
    csum = vec(np.ones(C_))
    d.call("msk_channel_sum", dt.msk(), vp(csum), 1)
    assert rel_err(vec_back(csum, C_), dout.astype(np.float64).sum(axis=(0, 2, 3, 4)) + 1.0) < 1e-5
    if C_ in (1, 2, 3, 6):
        # A/B: the scalar kernels (option dense12 0) give the same data gradients bit for bit (same arithmetic per element)
        d.set_option("dense12", 0)
        try:
            dx0, dr0, o0 = t_empty(N, C_, D, H, W), t_empty(N, C_, D, H, W, fill=1.0), t_empty(N, C_, D, H, W)
            d.call("msk_affine_act_bwd_apply", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), vp(sm), vp(si), vp(g), dt.msk(),
                   vp(sums), C.c_double(M), 1, dx0.msk(), dr0.msk(), 1)
            d.call("msk_affine_act_fwd", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), o0.msk())
        finally:
            d.set_option("dense12", 1)
        assert np.array_equal(t_to_ncdhw(dr0), t_to_ncdhw(drt)) and np.array_equal(t_to_ncdhw(o0), t_to_ncdhw(ot))
        assert rel_err(t_to_ncdhw(dx0), t_to_ncdhw(dxt)) < 1e-6
    # eval-mode coefficients
    d.call("msk_bn_eval_coeffs", C_, vp(g), vp(b_), vp(rmp), vp(rvp), C.c_float(1e-5), vp(sm), vp(si), vp(sc), vp(sh))
    d.call("msk_affine_act_fwd", xt.msk(), vp(sc), vp(sh), rt.msk(), vp(al), ot.msk())
    ev = O.prelu(O.bn_eval(x64, gamma, beta, vec_back(rmp, C_).astype(np.float64), vec_back(rvp, C_).astype(np.float64)) + res, alpha)
    assert rel_err(t_to_ncdhw(ot), ev) < 2e-5


def test_affine_act_tiled_residual_and_slices():
    """InputTransition's x.tile residual (res.c=1) and channel-slice outputs."""
    d = dev()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 16, 3, 4, 5)).astype(np.float32)
    r = rng.standard_normal((2, 1, 3, 4, 5)).astype(np.float32)
    alpha = rng.uniform(0.1, 0.4, 16).astype(np.float32)
    xt, rt = t_from_ncdhw(x), t_from_ncdhw(r)
    wide = t_empty(2, 32, 3, 4, 5, fill=0.0)
    d.call("msk_affine_act_fwd", xt.msk(), None, None, rt.msk(), vp(vec(alpha)), wide.channel_slice(16, 32).msk())
    got = wide.numpy()
    assert np.all(got[:, :16] == 0)
    assert rel_err(got[:, 16:], O.prelu((x + r).astype(np.float64), alpha.astype(np.float64))) < 1e-6


def test_copy_scale_channel_sum_argmax_layout():
    d = dev()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 20, 3, 5, 4)).astype(np.float32)
    xt = t_from_ncdhw(x)
    mask = (rng.random((2, 20)) < 0.5).astype(np.float32) * 2
    dst = t_empty(2, 20, 3, 5, 4, fill=1.0)
    d.call("msk_copy_scale", xt.msk(), vp(vec(mask.ravel())), dst.msk(), 1)
    assert rel_err(t_to_ncdhw(dst), x * mask[:, :, None, None, None] + 1.0) < 1e-6
    out = vec(np.ones(20))
    d.call("msk_channel_sum", xt.msk(), vp(out), 1)
    assert rel_err(vec_back(out, 20), x.sum(axis=(0, 2, 3, 4)) + 1.0) < 1e-5
    am = d.malloc(2 * 3 * 5 * 4 * 4)
    d.call("msk_argmax_c", xt.msk(), vp(am))
    assert np.array_equal(d.d2h(am, (2, 3, 5, 4), np.int32), x.argmax(axis=1))
    assert np.array_equal(xt.numpy(), x)  # NDHWC -> NCDHW round trip
    from medicalseg_amd.device import to_tensor
    assert np.array_equal(to_tensor(x).numpy(), x)  # NCDHW -> NDHWC -> NCDHW


@pytest.mark.parametrize("ncls,shape", [(3, (2, 6, 7, 8)), (20, (1, 5, 6, 12)), (2, (1, 4, 4, 4)),
                                        (20, (1, 9, 40, 50)), (8, (2, 5, 7, 11)), (12, (1, 6, 9, 10))])   # thread-per-voxel forms through the LDS tile: many tiles, ragged last tile, both register classes
def test_loss_fwd_bwd(ncls, shape):
    d = dev()
    N, D, H, W = shape
    rng = np.random.default_rng(ncls)
    z = (rng.standard_normal((N, ncls, D, H, W)) * 2).astype(np.float32)
    y = rng.integers(0, ncls, (N, D, H, W)).astype(np.int32)
    y[0, 0, 0, :2] = 255  # ignore_index voxels (CE only)
    zt = t_from_ncdhw(z)
    yp = d.malloc(y.nbytes)
    d.h2d(yp, y)
    wv = vec(np.zeros(ncls))
    d.call("msk_class_weights", zt.msk(), vp(wv))
    w_ref = O.class_weights(z.astype(np.float64))
    assert rel_err(vec_back(wv, ncls), w_ref) < 1e-5
    out, stats = vec(np.zeros(2 + ncls)), d.malloc((3 * ncls + 2) * 8)
    d.call("msk_loss_fwd", zt.msk(), vp(yp), vp(wv), 255, vp(out), vp(stats))
    ce_ref, dce = O.cross_entropy(z.astype(np.float64), y, w_ref, 255)
    ysafe = np.where(y == 255, 0, y)
    # dice of the oracle one-hots every voxel; mask the ignored ones out of t by hand
    C_ = ncls
    s = 1 / (1 + np.exp(-z.astype(np.float64)))
    t = np.moveaxis(np.eye(C_)[ysafe], -1, 1) * (y != 255)[:, None]
    inter, den = (s * t).sum((0, 2, 3, 4)), (s * s).sum((0, 2, 3, 4)) + (t * t).sum((0, 2, 3, 4))
    per = 2 * inter / np.maximum(den, 1e-6)
    o = vec_back(out, 2 + ncls)
    assert abs(o[0] - ce_ref) < 2e-5 * abs(ce_ref)
    assert abs(o[1] - (1 - per.mean())) < 2e-6
    assert rel_err(o[2:], per) < 2e-6
    dz = t_empty(N, ncls, D, H, W)
    d.call("msk_loss_bwd", zt.msk(), vp(yp), vp(wv), 255, vp(stats), C.c_float(0.7), C.c_float(1.3), dz.msk())
    ddice = -(1.0 / C_) * (2 * t / np.maximum(den, 1e-6).reshape(1, -1, 1, 1, 1)
                           - (2 * inter / np.maximum(den, 1e-6) ** 2).reshape(1, -1, 1, 1, 1) * 2 * s) * s * (1 - s)
    assert rel_err(t_to_ncdhw(dz), 0.7 * dce + 1.3 * ddice) < 2e-5


def test_loss_known_answer():
    """SURVEY.md Appendix C known-answer vector, through the device kernels."""
    d = dev()
    rng = np.random.default_rng(0)
    z = rng.standard_normal((1, 3, 2, 3, 4)).astype(np.float32)
    y = rng.integers(0, 3, (1, 2, 3, 4)).astype(np.int32)
    zt = t_from_ncdhw(z)
    yp = d.malloc(y.nbytes)
    d.h2d(yp, y)
    wv = vec(np.zeros(3))
    d.call("msk_class_weights", zt.msk(), vp(wv))
    assert np.allclose(vec_back(wv, 3), [2.47594326, 1.75639001, 1.86110799], rtol=2e-6)
    out, stats = vec(np.zeros(5)), d.malloc(11 * 8)
    d.call("msk_loss_fwd", zt.msk(), vp(yp), vp(wv), 255, vp(out), vp(stats))
    o = vec_back(out, 5)
    assert abs(o[0] - 1.4261519761109072) < 3e-6
    assert abs(o[1] - 0.5177017349991624) < 1e-6
    assert np.allclose(o[2:], [0.33488489, 0.69615749, 0.41585241], atol=1e-6)
    dz = t_empty(1, 3, 2, 3, 4)
    d.call("msk_loss_bwd", zt.msk(), vp(yp), vp(wv), 255, vp(stats), C.c_float(1.0), C.c_float(1.0), dz.msk())
    assert np.allclose(dz.numpy()[0, :, 0, 0, 0], [0.00738261, -0.0318339, 0.02601402], atol=2e-7)


def test_sgd_momentum():
    d = dev()
    rng = np.random.default_rng(3)
    n = 10007
    p, g, v = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
    pp, gp, vp_ = vec(p), vec(g), vec(v)
    d.call("msk_sgd_momentum", vp(pp), vp(gp), vp(vp_), C.c_size_t(n), C.c_float(0.01), C.c_float(0.9),
           C.c_float(1e-4), C.c_float(0.5))
    g2 = 0.5 * g.astype(np.float64) + 1e-4 * p
    v2 = 0.9 * v + g2
    assert rel_err(vec_back(vp_, n), v2) < 1e-6
    assert rel_err(vec_back(pp, n), p - 0.01 * v2) < 1e-6


def test_adam():
    """msk_adam (paddle.optimizer.Adam, cvlibs/config.py:214-216) over five steps against the float64 oracle."""
    d = dev()
    rng = np.random.default_rng(4)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32)
    pp, m1p, m2p = vec(p), vec(np.zeros(n)), vec(np.zeros(n))
    params, m1, m2 = {"p": p.astype(np.float64)}, {}, {}
    # the kernel holds beta1 / beta2 as fp32 like Paddle's (1 - float(0.999) differs from 0.001 by 4.7e-5 relative):
    # the oracle runs on the same rounded constants
    b1, b2 = float(np.float32(0.9)), float(np.float32(0.999))
    b1p, b2p = b1, b2
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        gp = vec(g)
        d.call("msk_adam", vp(pp), vp(gp), vp(m1p), vp(m2p), C.c_size_t(n), C.c_float(2e-3), C.c_float(0.9),
               C.c_float(0.999), C.c_float(1e-8), C.c_double(b1p), C.c_double(b2p), C.c_float(1e-4), C.c_float(0.5))
        b1p, b2p = b1p * b1, b2p * b2
        O.adam_step(params, {"p": 0.5 * g.astype(np.float64)}, m1, m2, t, 2e-3, b1, b2, 1e-8, float(np.float32(1e-4)))
        assert rel_err(vec_back(pp, n), params["p"]) < 1e-6
    assert rel_err(vec_back(m1p, n), m1["p"]) < 1e-6 and rel_err(vec_back(m2p, n), m2["p"]) < 1e-6


@pytest.mark.parametrize("sigmoid_norm,weighted", [(False, False), (True, True), (False, True)])
@pytest.mark.parametrize("ncls,shape", [(3, (2, 5, 6, 7)), (2, (1, 4, 8, 9)), (5, (1, 3, 4, 5))])
def test_dice_options(ncls, shape, sigmoid_norm, weighted):
    """DiceLoss(sigmoid_norm=False) and DiceLoss(weight=...) (dice_loss.py:36-43,68-69) through msk_loss_fwd_ex /
    msk_loss_bwd_ex against the float64 oracle; the CE term beside it is unchanged."""
    d = dev()
    N, D, H, W = shape
    rng = np.random.default_rng(ncls + 7)
    z = (rng.standard_normal((N, ncls, D, H, W)) * 2).astype(np.float32)
    y = rng.integers(0, ncls, (N, D, H, W)).astype(np.int32)
    zt = t_from_ncdhw(z)
    yp = d.malloc(y.nbytes)
    d.h2d(yp, y)
    w_ce = O.class_weights(z.astype(np.float64))
    wv = vec(w_ce)
    dw = rng.uniform(0.5, 2.0, ncls) if weighted else None
    dwp = vec(dw) if weighted else None
    out, stats = vec(np.zeros(2 + ncls)), d.malloc((3 * ncls + 2) * 8)
    d.call("msk_loss_fwd_ex", zt.msk(), vp(yp), vp(wv), 255, int(not sigmoid_norm), vp(dwp) if weighted else None,
           vp(out), vp(stats))
    z64 = z.astype(np.float64)
    ce_ref, dce = O.cross_entropy(z64, y, w_ce, 255)
    dl_ref, per_ref, ddl = O.dice(z64, y, sigmoid_norm=sigmoid_norm, weight=dw)
    o = vec_back(out, 2 + ncls)
    assert abs(o[0] - ce_ref) < 2e-5 * abs(ce_ref)
    assert abs(o[1] - dl_ref) < 2e-6 and rel_err(o[2:], per_ref) < 2e-6
    dz = t_empty(N, ncls, D, H, W)
    d.call("msk_loss_bwd_ex", zt.msk(), vp(yp), vp(wv), 255, int(not sigmoid_norm), vp(dwp) if weighted else None,
           vp(stats), C.c_float(0.7), C.c_float(1.3), dz.msk())
    assert rel_err(t_to_ncdhw(dz), 0.7 * dce + 1.3 * ddl) < 5e-6


def test_dropout_mask_statistics():
    d = dev()
    m = vec(np.zeros(4096))
    d.call("msk_dropout_mask", C.c_uint64(7), C.c_uint64(3), C.c_uint32(2), 4096, C.c_float(0.5), vp(m))
    a = vec_back(m, 4096)
    assert set(np.unique(a)) <= {0.0, 2.0}
    assert 0.45 < (a == 0).mean() < 0.55
    d.call("msk_dropout_mask", C.c_uint64(7), C.c_uint64(3), C.c_uint32(2), 4096, C.c_float(0.5), vp(m))
    assert np.array_equal(vec_back(m, 4096), a)  # deterministic in (seed, step, site)
    d.call("msk_dropout_mask", C.c_uint64(7), C.c_uint64(4), C.c_uint32(2), 4096, C.c_float(0.5), vp(m))
    assert not np.array_equal(vec_back(m, 4096), a)


INTERP_CASES = [
    # (C, (N, d, h, w) head resolution, (D, H, W) input resolution)
    (3, (2, 4, 4, 4), (32, 32, 32)),      # lung d1: x8 per axis
    (3, (1, 8, 8, 8), (32, 32, 32)),      # lung d2: x4
    (3, (1, 16, 16, 16), (32, 32, 32)),   # lung d3: x2
    (20, (1, 8, 8, 6), (64, 64, 12)),     # MRI d1: (8, 8, 2)
    (20, (1, 16, 16, 12), (32, 32, 12)),  # MRI d2/d3 class: last axis identity
    (5, (2, 5, 7, 3), (13, 9, 8)),        # ragged: non-integer ratios, one axis SHRINKS (7 -> 9, 5 -> 13)
    (4, (1, 6, 5, 4), (6, 5, 4)),         # identity
    (2, (1, 1, 1, 1), (5, 4, 3)),         # single source voxel
]


@pytest.mark.parametrize("case", INTERP_CASES)
def test_interp_trilinear_fwd_bwd(case):
    """msk_interp_trilinear_{fwd,bwd} vs the oracle's restatement of F.interpolate(mode='trilinear')
    (vnet_deepsup.py:268-277); 1e-5 of max|ref| (fp32 coordinates and weights vs float64)."""
    Cn, (N, sd, sh, sw), size = case
    d = dev()
    rng = np.random.default_rng(Cn * 1000 + sd * 7 + size[0])
    x = rng.standard_normal((N, Cn, sd, sh, sw)).astype(np.float32)
    y_ref = O.trilinear_resize(x.astype(np.float64), size)
    xt, yt = t_from_ncdhw(x), t_empty(N, Cn, *size, fill=9.0)
    d.call("msk_interp_trilinear_fwd", xt.msk(), yt.msk())
    assert rel_err(t_to_ncdhw(yt), y_ref) < 1e-5
    # a constant volume resizes to the same constant (weights sum to one)
    ct = t_from_ncdhw(np.full_like(x, 2.5))
    d.call("msk_interp_trilinear_fwd", ct.msk(), yt.msk())
    assert np.abs(t_to_ncdhw(yt) - 2.5).max() < 1e-5

    g = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.trilinear_resize_bwd(g.astype(np.float64), (sd, sh, sw))
    gt = t_from_ncdhw(g, ld=Cn + 3)  # gradient arriving as a channel slice of a wider buffer
    dxt = t_empty(N, Cn, sd, sh, sw, fill=1.0)
    need = C.c_size_t(0)
    d.call("msk_interp_scratch_bytes", dxt.msk(), gt.msk(), C.byref(need))
    scratch = d.malloc(max(need.value, 16))
    d.call("msk_interp_trilinear_bwd", gt.msk(), dxt.msk(), 0, vp(scratch), C.c_size_t(need.value))
    assert rel_err(t_to_ncdhw(dxt), dx_ref) < 1e-5
    d.call("msk_interp_trilinear_bwd", gt.msk(), dxt.msk(), 1, vp(scratch), C.c_size_t(need.value))
    assert rel_err(t_to_ncdhw(dxt), 2 * dx_ref) < 1e-5
    # adjoint identity <resize(x), g> == <x, resize^T(g)> with the device results
    d.call("msk_interp_trilinear_fwd", xt.msk(), yt.msk())
    d.call("msk_interp_trilinear_bwd", gt.msk(), dxt.msk(), 0, vp(scratch), C.c_size_t(need.value))
    lhs = float((t_to_ncdhw(yt).astype(np.float64) * g).sum())
    rhs = float((t_to_ncdhw(dxt).astype(np.float64) * x).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    if sh != size[1] or sw != size[2]:  # H/W passes need the scratch: too small must fail loudly
        from medicalseg_amd._lib import MskError
        with pytest.raises(MskError):
            d.call("msk_interp_trilinear_bwd", gt.msk(), dxt.msk(), 0, vp(scratch), C.c_size_t(4))


WINO_CASES = [
    # (Cin, Cout, (N, D, H, W)): whole 4x8x8 tiles; W % 16 == 0 -> F(4,5), else F(2,5)
    (32, 32, (2, 8, 16, 16)),
    (64, 40, (1, 4, 8, 32)),
    (16, 32, (1, 8, 8, 48)),
    (64, 48, (1, 4, 8, 24)),      # Cout not a multiple of 32
    (8, 40, (1, 12, 8, 8)),
    (128, 128, (1, 4, 16, 8)),    # W = 8: the transform runs along H (16) instead -> F(4,5) on permuted axes
    (16, 32, (1, 16, 32, 12)),    # MRI-like slab: W = 12 tiles along neither 8 nor 16 -> logical (d, h, w) = (W, D, H)
    (32, 20, (1, 8, 16, 4)),      # deeper MRI level (W = 4)
    (16, 16, (1, 8, 4, 16)),      # H = 4: logical (d, h, w) = (H, D, W)
    (32, 32, (1, 16, 16, 9)),     # MRI level 2 (W = 9): ragged planes along W (3 tiles of 4), transform along H
    (16, 24, (2, 6, 8, 16)),      # ragged D in the identity orientation
]


def _wino_variant(D, H, W):
    """Mirror of the axis choice in msk_gconv_halo_wino: any permutation of (D, H, W) with d % 4 == h % 8 == w % 8 == 0
    may carry the tiles; F(4,5) needs w % 16 == 0 and is preferred."""
    import itertools
    ok = [(d, h, w) for d, h, w in itertools.permutations((D, H, W))
          if h % 8 == 0 and w % 8 == 0 and (d + 3) // 4 * 4 * 2 <= d * 3]      # d may be ragged up to 1.5x padding
    assert ok
    return "conv_halo_wino4_k" if any(w % 16 == 0 for _, _, w in ok) else "conv_halo_wino_k"


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv5_winograd_f25_matches_oracle(case):
    """conv_halo_wino_k / conv_halo_wino4_k (1-D Winograd F(2,5) / F(4,5) along W, msk_conv_wino.hip) forward and data
    gradient vs the float64 oracle.  Tolerance: the conv tolerance of this file (8e-6 * sqrt(K/1000 + 1) of max|ref|);
    the transforms cost about one decimal digit relative to the direct fp32 kernel (measured ~1e-6 / ~4e-6 vs ~1e-7)."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(cin * 7 + cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y_ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s_, p)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(dy.astype(np.float64), w.astype(np.float64), x.shape, s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    dxt = t_empty(N, cin, D, H, W, fill=3.0)
    wp, bp = vec(w.ravel()), vec(b)
    d.set_option("conv_impl", 10)
    d.set_option("prof_shapes", 0)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        e_f, e_d = rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref)
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
        e_acc = rel_err(t_to_ncdhw(dxt), 2 * dx_ref)
        d.prof_enable(False)
        rep = d.prof_report()                                                # the Winograd kernel really ran:
        tag = _wino_variant(D, H, W)                                         # F(4,5) when some axis tiles by 16
        assert rep.get(tag, (0, 0))[0] == 3, rep
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    print("winograd rel err fwd %.2e dgrad %.2e" % (e_f, e_d))
    assert e_f < _conv_tol(cin * 125) and e_d < _conv_tol(cout * 125) and e_acc < _conv_tol(cout * 125)


def _wgrad_wino_variant(D, H, W):
    """Mirror of msk_wgrad_wino's axis choice: the transform axis is any axis that is even and >= 16; F(4,5) when one
    of them is a multiple of 4."""
    ws = [w for w in (D, H, W) if w % 2 == 0 and w >= 8]
    assert ws
    return "wgrad_wino4" if any(w % 4 == 0 for w in ws) else "wgrad_wino"


@pytest.mark.parametrize("case", [(32, 32, (2, 8, 16, 16)), (64, 48, (1, 4, 8, 24)), (8, 40, (1, 5, 9, 18)),
                                  (128, 128, (1, 4, 16, 16)),
                                  (16, 32, (1, 16, 32, 12)),    # MRI-like slab: transform along H, planes along W
                                  (32, 16, (1, 20, 24, 9)),     # odd W (MRI level 2)
                                  (16, 16, (1, 32, 18, 2)),     # W = 2 (deepest MRI level); 18: F(2,5) unless D (32) wins
                                  (32, 32, (2, 8, 8, 8)),       # 8^3 (deepest lung level): half-empty 16-column chunks
                                  (8, 8, (1, 6, 18, 3))])       # only H is usable and 18 % 4 != 0 -> F(2,5) along H
def test_wgrad_winograd_f25_matches_oracle(case):
    """wgrad_wino_k (msk_wgrad_wino.hip): the adjoint of the Winograd forward kernel, dU accumulated in the transformed
    domain and mapped back with G^T in the split-K reduction.  Same tolerance as the direct weight-gradient kernels."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(cin * 11 + cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    dw_ref, db_ref = O.conv3d_wgrad(dy.astype(np.float64), x.astype(np.float64), k, s_, p)
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    dwp, dbp = vec(np.full(cout * cin * 125, 0.5, np.float32)), vec(np.zeros(cout, np.float32))
    d.set_option("conv_impl", 12)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
        got = vec_back(dwp, dw_ref.size).reshape(dw_ref.shape)
        d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 1)
        got2 = vec_back(dwp, dw_ref.size).reshape(dw_ref.shape)
        d.prof_enable(False)
        assert d.prof_report().get(_wgrad_wino_variant(D, H, W), (0, 0))[0] == 2                   # F(4,5) / F(2,5)
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    M = N * D * H * W
    e1, e2 = rel_err(got, dw_ref), rel_err(got2, 2 * dw_ref)
    print("winograd wgrad rel err %.2e" % e1)
    assert e1 < _conv_tol(M) * 2 and e2 < _conv_tol(M) * 2


FOLDN_CASES = [
    # (Cout, (N, D, H, W)): out_tr.conv1 class (32 -> ncls <= 3, 5^3 same) on conv_foldn_k
    (3, (2, 20, 13, 37)),     # ragged (h, w) tiles, D split into segments
    (3, (1, 4, 8, 16)),       # one tile, D at the minimum
    (2, (1, 33, 8, 12)),      # narrowest W, odd D
    (1, (2, 9, 24, 48)),
    (3, (1, 64, 16, 16)),     # long march: many planes per workgroup
]


@pytest.mark.parametrize("case", FOLDN_CASES)
def test_conv_foldn_out_tr_class(case):
    """conv_foldn_k (kd taps folded into the MFMA columns, marching along D; msk_conv_foldn.hip) against the float64
    oracle and against the VALU kernel it replaced (conv_impl 22), forward and accumulate... the data gradient of this
    class never routes here (CK = ncls)."""
    cout, (N, D, H, W) = case
    d = dev()
    rng = np.random.default_rng(100 * cout + D)
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    x = rng.standard_normal((N, 32, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, 32) + k) / np.sqrt(32 * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s_, p)
    xt = t_from_ncdhw(x)
    wp, bp = vec(w.ravel()), vec(b)
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    out = {}
    try:
        for impl in (0, 22, 24):             # product (fp16 two-piece MFMA), VALU kernel, fp32-MFMA form of the folded kernel
            d.set_option("conv_impl", impl)
            yt = t_empty(N, cout, D, H, W, fill=7.0)
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
            out[impl] = t_to_ncdhw(yt)
            d.prof_enable(False)
            rep = d.prof_report()
            assert any(k.startswith("conv_foldn") for k in rep) == (impl != 22), rep
            assert ("conv_foldn_h2" in rep) == (impl == 0), rep
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    e0, e22, e24 = rel_err(out[0], ref), rel_err(out[22], ref), rel_err(out[24], ref)
    print("foldn fp16x2 %.2e  fp32-MFMA %.2e  valu %.2e" % (e0, e24, e22))
    assert e0 < _conv_tol(32 * 125) and e22 < _conv_tol(32 * 125) and e24 < _conv_tol(32 * 125)
    # a strided destination (channel slice of a wider tensor) keeps its neighbours
    wide = t_empty(N, cout + 2, D, H, W, fill=5.0)
    d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), wide.channel_slice(1, 1 + cout).msk())
    got = wide.numpy()
    assert rel_err(got[:, 1:1 + cout], ref) < _conv_tol(32 * 125)
    assert np.all(got[:, 0] == 5.0) and np.all(got[:, -1] == 5.0)


@pytest.mark.parametrize("case", [(5, 16, (2, 9, 13, 37)), (5, 16, (1, 4, 8, 32)), (5, 12, (1, 6, 8, 16)), (5, 5, (2, 5, 9, 40)),
                                  (3, 32, (2, 9, 13, 37)), (3, 16, (1, 6, 8, 33)), (3, 22, (1, 5, 9, 40))])
def test_conv_c1_in_tr_class(case):
    """One input channel, 5^3 -> <= 16 channels (in_tr.conv1, vnet.py:67) or 3^3 -> <= 32 (first convolution of the builder-defined
    UNet3D) against the float64 oracle: conv_c1_h2_k (round 4: 16-bit matrix pipe, fp16 x 2 operand pieces -- the product), the
    fp32-MFMA kernel it replaced (conv_impl 27, 5^3 only) and the VALU / general kernels (conv_impl 23), incl. output-channel
    counts that are not a multiple of 4 and a strided destination; and the single-fp16 form (conv_fp16, 3^3) at the stated
    fp16 tolerance."""
    ks, cout, (N, D, H, W) = case
    d = dev()
    rng = np.random.default_rng(cout + D + ks)
    k, s_, p = (ks,) * 3, (1, 1, 1), (ks // 2,) * 3
    x = (rng.standard_normal((N, 1, D, H, W)) * 3.0 + 0.7).astype(np.float32)
    w = (rng.standard_normal((cout, 1) + k) / np.sqrt(ks ** 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s_, p)
    xt, wp, bp = t_from_ncdhw(x), vec(w.ravel()), vec(b)
    ld = (cout + 3) // 4 * 4 + 4
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    d.set_option("c1_h2", 2)     # the product default (1) keeps the fp32-MFMA kernel for 5^3: cover both instantiations here
    try:
        for impl, fp16 in ((0, 0), (27, 0), (23, 0), (0, 1)):
            if fp16 and ks != 3:
                continue
            d.set_option("conv_impl", impl)
            d.set_option("conv_fp16", fp16)
            yt = t_empty(N, cout, D, H, W, ld=ld, fill=7.0)
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
            got = yt.numpy()
            d.prof_enable(False)
            rep = d.prof_report()
            assert ("conv_c1_h2" in rep) == (impl == 0), rep
            if ks == 5:
                assert ("conv_c1_mfma" in rep) == (impl == 27), rep
            e = rel_err(got, ref)
            print("conv c1 k=%d cout=%d impl %d fp16 %d: %.2e" % (ks, cout, impl, fp16, e))
            assert e < (3e-3 if fp16 else _conv_tol(ks ** 3)), (impl, fp16, e)
            full = d.d2h(yt.ptr, (N, D, H, W, ld), np.float32)
            assert np.all(full[..., cout:] == 7.0)          # the padding channels of the strided destination are untouched
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
        d.set_option("conv_fp16", 0)
        d.set_option("c1_h2", 1)


FOLD_CASES = [
    # (Cin, Cout, k, pad, (N, D, H, W), kernel that must run)
    (32, 32, 5, 2, (2, 8, 16, 16), "conv_halo_wino4_k"),      # F(4,5) epilogue (fp32-MFMA kernels: wino_bf3 = 0 below)
    (32, 32, 5, 2, (2, 16, 16, 16), "wbf_tout_k"),            # bf16x3 pipeline: slope applied by the output transform
    (64, 48, 5, 2, (1, 4, 8, 24), "conv_halo_wino_k"),        # F(2,5) epilogue, Cout not a multiple of 32
    (32, 24, 5, 2, (1, 16, 32, 12), "conv_halo_wino4_k"),     # permuted axes (transform along H)
    (128, 128, 5, 2, (1, 4, 8, 16), "conv_splitk_reduce"),    # few tiles -> split K: slope applied in the reduce
    (16, 8, 3, 1, (1, 6, 7, 9), "prelu_inplace"),             # no Winograd kernel: in-place pass after the conv
    (32, 3, 5, 2, (1, 9, 11, 21), "conv_foldn_h2"),           # out_tr.conv1: slope in the folded-column MFMA kernel's epilogue
]


@pytest.mark.parametrize("case", FOLD_CASES)
def test_conv_fold_bn_and_fused_prelu_epilogue(case):
    """Inference path: msk_conv_fold_bn + msk_conv3d_fwd_act == PReLU(BN_eval(conv(x))) of the float64
    oracle, through every epilogue that applies the slope (both Winograd kernels, the split-K reduce) and through the
    in-place pass behind the other kernels."""
    cin, cout, k_, p_, (N, D, H, W), tag = case
    k, s_, p = (k_,) * 3, (1, 1, 1), (p_,) * 3
    d = dev()
    rng = np.random.default_rng(cin + 3 * cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * k_ ** 3)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    gamma, beta = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    mean, var = rng.standard_normal(cout).astype(np.float32), rng.uniform(0.5, 2.0, cout).astype(np.float32)
    slope = rng.uniform(0.05, 0.5, cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    sh = (1, cout, 1, 1, 1)
    z = (y - f8(mean).reshape(sh)) / np.sqrt(f8(var).reshape(sh) + 1e-5) * f8(gamma).reshape(sh) + f8(beta).reshape(sh)
    ref = np.where(z > 0, z, z * f8(slope).reshape(sh))
    xt, yt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0)
    wp, bp, wf, bf = vec(w.ravel()), vec(b), vec(np.zeros(w.size)), vec(np.zeros(cout))
    sc = [vec(np.zeros(cout)) for _ in range(4)]   # mean, invstd, scale, shift
    d.call("msk_bn_eval_coeffs", cout, vp(vec(gamma)), vp(vec(beta)), vp(vec(mean)), vp(vec(var)), C.c_float(1e-5),
           vp(sc[0]), vp(sc[1]), vp(sc[2]), vp(sc[3]))
    d.call("msk_conv_fold_bn", vp(wp), vp(bp), vp(sc[2]), vp(sc[3]), cout, C.c_long(w.size // cout), vp(wf), vp(bf))
    scale = f8(gamma) / np.sqrt(f8(var) + 1e-5)
    assert rel_err(vec_back(wf, w.size).reshape(w.shape), f8(w) * scale.reshape(cout, 1, 1, 1, 1)) < 1e-6
    assert rel_err(vec_back(bf, cout), f8(b) * scale + f8(beta) - f8(mean) * scale) < 1e-6
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    d.set_option("wino_bf3", 1 if tag.startswith("wbf") else 0)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd_act", _desc(k, s_, p), xt.msk(), vp(wf), vp(bf), vp(vec(slope)), yt.msk())
        got = t_to_ncdhw(yt)
        d.prof_enable(False)
        rep = d.prof_report()
        assert rep.get(tag, (0, 0))[0] == 1, rep
        assert ("prelu_inplace" in rep) == (tag == "prelu_inplace"), rep
    finally:
        d.prof_enable(False)
        d.set_option("wino_bf3", 1)
    e = rel_err(got, ref)
    print("folded conv rel err %.2e" % e)
    assert e < _conv_tol(cin * k_ ** 3)
    # slope NULL: plain convolution with the folded weights
    d.call("msk_conv3d_fwd_act", _desc(k, s_, p), xt.msk(), vp(wf), vp(bf), None, yt.msk())
    assert rel_err(t_to_ncdhw(yt), z) < _conv_tol(cin * k_ ** 3)


def _fuzz_cases():
    rng = np.random.default_rng(20260928)
    sizes = [2, 3, 4, 5, 8, 9, 12, 16, 18, 24, 32]
    chans = [4, 8, 12, 16, 20, 32, 40]
    cases = []
    while len(cases) < 24:
        n = int(rng.integers(1, 3))
        d, h, w = (int(rng.choice(sizes)) for _ in range(3))
        if n * d * h * w > 9000:
            continue
        cases.append((int(rng.choice(chans)), int(rng.choice(chans)), (n, d, h, w)))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases())
def test_conv5_dispatch_fuzz_matches_oracle(case):
    """Seeded shape fuzz of the 5^3 'same' convolution through the DEFAULT dispatch (Winograd on whichever axis
    permutation tiles, ragged planes, split-K, direct MFMA / VALU fallbacks): forward, data gradient (fresh and
    accumulating) and weight gradient against the float64 oracle, whatever kernel the shape lands on."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    d = dev()
    rng = np.random.default_rng(cin * 131 + cout * 17 + D * 7 + H * 3 + W)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 125)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = rng.standard_normal((N, cout, D, H, W)).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    dw_ref, db_ref = O.conv3d_wgrad(f8(dy), f8(x), k, s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    dxt = t_empty(N, cin, D, H, W, fill=3.0)
    wp, bp = vec(w.ravel()), vec(b)
    dwp, dbp = vec(np.full(w.size, 0.25, np.float32)), vec(np.zeros(cout, np.float32))
    d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
    d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
    e_f, e_d = rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref)
    d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
    e_acc = rel_err(t_to_ncdhw(dxt), 2 * dx_ref)
    d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
    e_w = rel_err(vec_back(dwp, w.size).reshape(w.shape), dw_ref)
    e_b = rel_err(vec_back(dbp, cout), db_ref)
    M = N * D * H * W
    assert e_f < _conv_tol(cin * 125) and e_d < _conv_tol(cout * 125) and e_acc < _conv_tol(cout * 125), (e_f, e_d, e_acc)
    assert e_w < _conv_tol(M) * 2 and e_b < 1e-5, (e_w, e_b)


WINO3_CASES = [
    # (Cin, Cout, (N, D, H, W)): 3 x 3 x 3 'same' convs that tile 4 x 8 x 16 over some axis permutation -> F(4,3)
    (32, 32, (2, 8, 16, 16)),
    (64, 40, (1, 4, 8, 32)),       # Cout not a multiple of 32
    (16, 24, (1, 6, 8, 16)),       # ragged planes (6 = 4 + 2)
    (16, 32, (1, 16, 32, 12)),     # permuted axes (transform along H)
    (128, 128, (1, 4, 8, 16)),     # one tile -> split K
    (12, 20, (2, 6, 8, 48)),       # Cin not a multiple of 8, ragged D
]


@pytest.mark.parametrize("case", WINO3_CASES)
def test_conv3_winograd_f43_matches_oracle(case):
    """conv_halo_wino43_k (1-D Winograd F(4,3) for the 3 x 3 x 3 convolutions: UNet3D's DoubleConvs, the deep-supervision
    heads) forward, data gradient (fresh and accumulating) against the float64 oracle at the conv tolerance of this file."""
    cin, cout, (N, D, H, W) = case
    k, s_, p = (3, 3, 3), (1, 1, 1), (1, 1, 1)
    d = dev()
    rng = np.random.default_rng(cin * 5 + cout)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin) + k) / np.sqrt(cin * 27)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    f8 = lambda a: a.astype(np.float64)
    y_ref = O.conv3d(f8(x), f8(w), f8(b), s_, p)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p)
    xt, yt, dyt = t_from_ncdhw(x), t_empty(N, cout, D, H, W, fill=7.0), t_from_ncdhw(dy)
    dxt = t_empty(N, cin, D, H, W, fill=3.0)
    wp, bp = vec(w.ravel()), vec(b)
    d.set_option("conv_impl", 10)
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    try:
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_conv3d_fwd", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk())
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
        e_f, e_d = rel_err(t_to_ncdhw(yt), y_ref), rel_err(t_to_ncdhw(dxt), dx_ref)
        d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
        e_acc = rel_err(t_to_ncdhw(dxt), 2 * dx_ref)
        d.prof_enable(False)
        assert d.prof_report().get("conv_halo_wino43_k", (0, 0))[0] == 3, d.prof_report()
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    print("F(4,3) rel err fwd %.2e dgrad %.2e" % (e_f, e_d))
    assert e_f < _conv_tol(cin * 27) and e_d < _conv_tol(cout * 27) and e_acc < _conv_tol(cout * 27), (e_f, e_d, e_acc)


@pytest.mark.parametrize("shape", [(2, 32, 4, 6, 10), (1, 64, 3, 5, 7), (1, 12, 4, 4, 5)])
def test_affine_act_join_fwd_bwd(shape):
    """msk_affine_act_join_fwd / msk_add_act_join_bwd: the residual join fused with the BatchNorm apply + PReLU of the unit
    in front of it (vnet.py:107-111, 150-154), against numpy float64 of the two-step form
        a = prelu(scale*y + shift, alpha_in);  out = prelu(a + res, alpha_out)
    and its adjoint (da, dres (+)=, dalpha_out +=)."""
    from medicalseg_amd._lib import NULL_TENSOR  # noqa: F401
    N, Cn, D, H, W = shape
    d = dev()
    rng = np.random.default_rng(Cn)
    f8 = lambda a: a.astype(np.float64)
    y = rng.standard_normal(shape).astype(np.float32)
    res = rng.standard_normal(shape).astype(np.float32)
    dout = rng.standard_normal(shape).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, Cn).astype(np.float32), rng.standard_normal(Cn).astype(np.float32)
    ai, ao = rng.uniform(0.05, 0.5, Cn).astype(np.float32), rng.uniform(-0.2, 0.5, Cn).astype(np.float32)
    sh = (1, Cn, 1, 1, 1)
    u = f8(y) * f8(scale).reshape(sh) + f8(shift).reshape(sh)
    a = np.where(u > 0, u, u * f8(ai).reshape(sh))
    s_ = a + f8(res)
    out_ref = np.where(s_ > 0, s_, s_ * f8(ao).reshape(sh))
    ds = f8(dout) * np.where(s_ > 0, 1.0, f8(ao).reshape(sh))
    dao_ref = (f8(dout) * s_ * (s_ <= 0)).sum(axis=(0, 2, 3, 4))
    yt, rt, dt = t_from_ncdhw(y), t_from_ncdhw(res), t_from_ncdhw(dout)
    ot = t_empty(N, Cn, D, H, W, fill=7.0)
    sc, sf, pai, pao = vec(scale), vec(shift), vec(ai), vec(ao)
    d.call("msk_affine_act_join_fwd", yt.msk(), vp(sc), vp(sf), vp(pai), rt.msk(), vp(pao), ot.msk())
    assert rel_err(t_to_ncdhw(ot), out_ref) < 1e-6
    da, dres = t_empty(N, Cn, D, H, W, fill=5.0), t_from_ncdhw(np.full(shape, 0.5, np.float32))
    dao = vec(np.full(Cn, 0.25, np.float32))
    d.call("msk_add_act_join_bwd", yt.msk(), vp(sc), vp(sf), vp(pai), rt.msk(), vp(pao), dt.msk(), da.msk(), dres.msk(), 1, vp(dao))
    assert rel_err(t_to_ncdhw(da), ds) < 1e-6
    assert rel_err(t_to_ncdhw(dres), ds + 0.5) < 1e-6
    assert rel_err(vec_back(dao, Cn), dao_ref + 0.25) < 2e-5
    # the _ex form: the same outputs plus the unit's own backward sums (what msk_affine_act_bwd_reduce_ex(y, dout = da) gives)
    if Cn % 4 == 0:
        mean, invstd = rng.standard_normal(Cn).astype(np.float32), rng.uniform(0.5, 2.0, Cn).astype(np.float32)
        da2, dres2 = t_empty(N, Cn, D, H, W, fill=5.0), t_from_ncdhw(np.full(shape, 0.5, np.float32))
        dao2, sums4, maxes = vec(np.full(Cn, 0.25, np.float32)), vec(np.zeros(4 * Cn, np.float32)), vec(np.zeros(128, np.float32))
        d.call("msk_add_act_join_bwd_ex", yt.msk(), vp(sc), vp(sf), vp(pai), rt.msk(), vp(pao), vp(vec(mean)), vp(vec(invstd)),
               dt.msk(), da2.msk(), dres2.msk(), 1, vp(dao2), vp(sums4), vp(maxes))
        assert rel_err(t_to_ncdhw(da2), ds) < 1e-6 and rel_err(t_to_ncdhw(dres2), ds + 0.5) < 1e-6
        assert rel_err(vec_back(dao2, Cn), dao_ref + 0.25) < 2e-5
        du = ds * np.where(u > 0, 1.0, f8(ai).reshape(sh))
        xhat = (f8(y) - f8(mean).reshape(sh)) * f8(invstd).reshape(sh)
        ref_sums = np.concatenate([du.sum(axis=(0, 2, 3, 4)), (du * xhat).sum(axis=(0, 2, 3, 4)), (ds * u * (u <= 0)).sum(axis=(0, 2, 3, 4))])
        got = vec_back(sums4, 3 * Cn)
        assert np.abs(got - ref_sums).max() < 2e-5 * np.abs(ref_sums).max()
        mx = d.d2h(maxes, (2, 64), np.float32).max(axis=1)
        assert abs(mx[0] - np.abs(du).max()) < 1e-6 * np.abs(du).max() and abs(mx[1] - np.abs(xhat).max()) < 1e-5 * np.abs(xhat).max()



@pytest.mark.parametrize("case", [(3, (2, 20, 16, 37)), (3, (1, 4, 8, 12)), (2, (1, 33, 24, 16)), (4, (1, 9, 17, 48)), (1, (2, 8, 8, 16))])
def test_conv_tk_h2_out_tr_dgrad_class(case):
    """conv_tk_h2_k (data gradient of out_tr.conv1, ncls <= 4 -> 32 channels, vnet.py:165: fp16 two-piece operands, kd folded
    into the MFMA K dimension, ring of dy planes in LDS, marching along D) against the float64 oracle and against the fp32-MFMA
    tight-K kernel (conv_impl 25), plain and accumulating, gradient magnitudes 1e-6."""
    ncls, (N, D, H, W) = case
    d = dev()
    rng = np.random.default_rng(50 * ncls + D)
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    w = (rng.standard_normal((ncls, 32) + k) / np.sqrt(32 * 125)).astype(np.float32)
    dy = (rng.standard_normal((N, ncls, D, H, W)) * 1e-6).astype(np.float32)
    ref = O.conv3d_dgrad(dy.astype(np.float64), w.astype(np.float64), (N, 32, D, H, W), s_, p)
    dyt, wp = t_from_ncdhw(dy), vec(w.ravel())
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    out = {}
    try:
        for impl in (0, 25):
            d.set_option("conv_impl", impl)
            dxt = t_empty(N, 32, D, H, W, fill=3.0)
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 0)
            out[impl] = t_to_ncdhw(dxt)
            d.call("msk_conv3d_dgrad", _desc(k, s_, p), dyt.msk(), vp(wp), dxt.msk(), 1)
            acc = t_to_ncdhw(dxt)
            d.prof_enable(False)
            rep = d.prof_report()
            assert ("conv_tk_h2" in rep) == (impl == 0), rep
            assert rel_err(acc, 2 * ref) < _conv_tol(ncls * 125)
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    e0, e25 = rel_err(out[0], ref), rel_err(out[25], ref)
    print("tk fp16x2 %.2e  fp32-MFMA %.2e" % (e0, e25))
    assert e0 < _conv_tol(ncls * 125) and e25 < _conv_tol(ncls * 125)


@pytest.mark.parametrize("case", [(3, 32, (2, 20, 16, 37)), (3, 32, (1, 4, 8, 12)), (2, 16, (1, 9, 24, 40)), (4, 32, (1, 9, 17, 48)),
                                  (1, 8, (2, 8, 8, 16)), (3, 24, (1, 6, 10, 33))])
def test_wgrad_cbs_h2_out_tr_class(case):
    """wgrad_cbs_h2_k (weight gradient of out_tr.conv1, 32 -> ncls <= 4, vnet.py:165: fp16 two-piece operands, 32 voxels per
    MFMA, dy as shifted {v, v+1} dword pairs in LDS) against the float64 oracle and against the fp32-MFMA kernel
    (conv_impl 26), plain and accumulating; dy at gradient magnitude 1e-6, x at O(1)."""
    ncls, cin, (N, D, H, W) = case
    d = dev()
    rng = np.random.default_rng(60 * ncls + D)
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    dy = (rng.standard_normal((N, ncls, D, H, W)) * 1e-6).astype(np.float32)
    ref = O.conv3d_wgrad(dy.astype(np.float64), x.astype(np.float64), k, s_, p)[0]
    xt, dyt = t_from_ncdhw(x), t_from_ncdhw(dy)
    M = N * D * H * W
    d.set_option("prof_shapes", 0)
    d.set_option("prof_only_halo", 0)
    out = {}
    try:
        for impl in (0, 26):
            d.set_option("conv_impl", impl)
            dwp, dbp = vec(np.full(ref.size, 0.5, np.float32)), vec(np.zeros(ncls))
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
            out[impl] = vec_back(dwp, ref.size).reshape(ref.shape)
            d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 1)
            acc = vec_back(dwp, ref.size).reshape(ref.shape)
            d.prof_enable(False)
            rep = d.prof_report()
            h2 = impl == 0 and ncls <= 3   # four classes = 32 row tiles = 256 accumulator registers: fp32 kernel
            assert ("wgrad_cbs_h2" in rep) == h2 and ("wgrad_cbs_mfma" in rep) == (not h2), rep
            assert rel_err(acc, 2 * ref) < _conv_tol(M) * 2
    finally:
        d.prof_enable(False)
        d.set_option("conv_impl", 0)
    e0, e26 = rel_err(out[0], ref), rel_err(out[26], ref)
    print("cbs fp16x2 %.2e  fp32-MFMA %.2e" % (e0, e26))
    assert e0 < _conv_tol(M) * 2 and e26 < _conv_tol(M) * 2


def test_elu_fwd_bwd():
    """msk_elu_fwd / msk_elu_bwd (paddle.nn.ELU, vnet.py:25-29): values, in-place forward, channel-slice views,
    the derivative taken from the output, accumulate."""
    d = dev()
    rng = np.random.default_rng(12)
    N, Cc, D, H, W = 2, 6, 3, 5, 7
    x = (rng.standard_normal((N, Cc, D, H, W)) * 2).astype(np.float32)
    x[0, 0, 0, 0, :3] = [0.0, -0.0, 1e-30]
    dout = rng.standard_normal(x.shape).astype(np.float32)
    for alpha in (1.0, 0.5):
        ref = np.where(x > 0, x, alpha * np.expm1(np.minimum(x.astype(np.float64), 0)))
        dref = dout * np.where(x > 0, 1.0, alpha * np.exp(np.minimum(x.astype(np.float64), 0)))
        xt, ot = t_from_ncdhw(x), t_empty(N, Cc, D, H, W, fill=9.0)
        d.call("msk_elu_fwd", xt.msk(), C.c_float(alpha), ot.msk())
        assert rel_err(t_to_ncdhw(ot), ref) < 1e-6
        d.call("msk_elu_fwd", xt.msk(), C.c_float(alpha), xt.msk())          # in place
        assert np.array_equal(t_to_ncdhw(xt), t_to_ncdhw(ot))
        dt, gt = t_from_ncdhw(dout), t_empty(N, Cc, D, H, W, fill=1.0)
        d.call("msk_elu_bwd", ot.msk(), dt.msk(), C.c_float(alpha), gt.msk(), 0)
        assert rel_err(t_to_ncdhw(gt), dref) < 2e-6
        d.call("msk_elu_bwd", ot.msk(), dt.msk(), C.c_float(alpha), gt.msk(), 1)
        assert rel_err(t_to_ncdhw(gt), 2 * dref) < 2e-6


def test_out_tr_amax_travels_in_xform_header():
    """out_tr.conv1 class (32 -> 3, vnet.py:165): msk_conv3d_xform_bytes is the 512-byte header, the forward kernel leaves max |x|
    there, and the weight gradient that takes it (no absmax pass over x) equals the one that measures x itself -- bitwise,
    both scale by the same power of two."""
    d = dev()
    rng = np.random.default_rng(77)
    N, D, H, W = 2, 8, 16, 24
    k, s_, p = (5, 5, 5), (1, 1, 1), (2, 2, 2)
    x = rng.standard_normal((N, 32, D, H, W)).astype(np.float32) * 3.7
    w = (rng.standard_normal((3, 32) + k) / 60).astype(np.float32)
    b = rng.standard_normal(3).astype(np.float32)
    dy = (rng.standard_normal((N, 3, D, H, W)) * 1e-4).astype(np.float32)
    xt, dyt, wp, bp = t_from_ncdhw(x), t_from_ncdhw(dy), vec(w.ravel()), vec(b)
    nbytes = int(d.lib.msk_conv3d_xform_bytes(d.ctx, _desc(k, s_, p), xt.msk(), 3))
    assert nbytes == 512
    xf = d.malloc(nbytes)
    yt = t_empty(N, 3, D, H, W)
    d.call("msk_conv3d_fwd_ex", _desc(k, s_, p), xt.msk(), vp(wp), vp(bp), yt.msk(), None, C.c_void_p(xf))
    amax = d.d2h(xf, (64,), np.float32)
    assert amax.max() == np.abs(x).max()
    y_ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s_, p)
    assert rel_err(t_to_ncdhw(yt), y_ref) < _conv_tol(32 * 125)
    d.set_option("prof_only_halo", 0)
    outs = []
    try:
        for use in (True, False):
            dwp, dbp = vec(np.zeros(w.size, np.float32)), vec(np.zeros(3))
            d.prof_reset()
            d.prof_enable(True)
            if use:
                d.call("msk_conv3d_wgrad_ex", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0, C.c_void_p(xf))
            else:
                d.call("msk_conv3d_wgrad", _desc(k, s_, p), xt.msk(), dyt.msk(), vp(dwp), vp(dbp), 0)
            d.prof_enable(False)
            rep = d.prof_report()
            assert "wgrad_cbs_h2" in rep and rep["absmax"][0] == (1 if use else 2), rep
            outs.append(vec_back(dwp, w.size))
    finally:
        d.prof_enable(False)
    assert np.array_equal(outs[0], outs[1])
    ref = O.conv3d_wgrad(dy.astype(np.float64), x.astype(np.float64), k, s_, p)[0]
    assert rel_err(outs[0].reshape(ref.shape), ref) < 2 * _conv_tol(N * D * H * W)


def test_tile_staged_records_equal_the_direct_form_bitwise():
    """Option "tile_staging" (round 4): the dense 20-class head (pointwise_mid) and its loss kernels move their voxel records
    through an LDS tile with whole-line accesses; the arithmetic per voxel and the order of every sum are those of the
    one-thread-per-voxel form reading HBM directly, so the results are the same bits."""
    d = dev()
    ncls, (N, D, H, W) = 20, (1, 7, 33, 41)
    rng = np.random.default_rng(77)
    z = (rng.standard_normal((N, ncls, D, H, W)) * 2).astype(np.float32)
    y = rng.integers(0, ncls, (N, D, H, W)).astype(np.int32)
    w = (rng.standard_normal((ncls, ncls, 1, 1, 1)) / np.sqrt(ncls)).astype(np.float32)
    b = rng.standard_normal(ncls).astype(np.float32)
    from medicalseg_amd._lib import MskConvDesc
    cd = MskConvDesc(1, 1, 1, 1, 1, 1, 0, 0, 0)
    zt = t_from_ncdhw(z)
    yp = d.malloc(y.nbytes)
    d.h2d(yp, y)
    wp, bp = vec(w.ravel()), vec(b)
    res = {}
    try:
        for mode in (1, 0):
            d.set_option("tile_staging", 7 if mode else 0)
            lt, acc = t_empty(N, ncls, D, H, W, fill=5.0), t_from_ncdhw(z)
            d.prof_reset()
            d.prof_enable(True)
            d.call("msk_conv3d_fwd", cd, zt.msk(), vp(wp), vp(bp), lt.msk())
            d.call("msk_conv3d_dgrad", cd, zt.msk(), vp(wp), acc.msk(), 1)       # accumulating store pass
            d.prof_enable(False)
            assert d.prof_report().get("pointwise_mid", (0, 0))[0] == 2
            wv = vec(np.ones(ncls))
            out, stats = vec(np.zeros(2 + ncls)), d.malloc((3 * ncls + 2) * 8)
            d.call("msk_loss_fwd", lt.msk(), vp(yp), vp(wv), 255, vp(out), vp(stats))
            dz = t_empty(N, ncls, D, H, W, fill=9.0)
            d.call("msk_loss_bwd", lt.msk(), vp(yp), vp(wv), 255, vp(stats), C.c_float(1.0), C.c_float(1.0), dz.msk())
            res[mode] = (lt.numpy(), acc.numpy(), vec_back(out, 2 + ncls), dz.numpy())
    finally:
        d.set_option("tile_staging", 7)
    for a, b_ in zip(res[1], res[0]):
        assert np.array_equal(a, b_)
    ref = O.conv3d(z.astype(np.float64), w.astype(np.float64), b.astype(np.float64), (1, 1, 1), (0, 0, 0))
    assert rel_err(res[1][0], ref) < 1e-5


def test_small_pack_cache_follows_every_weight_write():
    """Round 5: the packed weight images of the kernel == stride convolutions (forward gather, data-gradient scatter) and of the
    1x1x1 head are cached per (weight tensor, layout) and rebuilt in one launch by the optimizer kernels (msk_conv.hip
    SmallPackCache).  Every entry point that can change the tensor must invalidate them: h2d, d2d, memset, the optimizer
    kernels, free + reuse; option small_pack_cache 0 = the per-call packs (same results)."""
    d = dev()
    rng = np.random.default_rng(11)
    f8 = lambda a: a.astype(np.float64)
    cases = [(16, 32, (2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 32, 32, 32)),     # gconv_ks_fwd / convT_scatter (>= 16384 source voxels)
             (20, 20, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 8, 16, 16))]      # pointwise_mid
    for cin, cout, k, s_, p, (N, D, H, W) in cases:
        x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
        Do, Ho, Wo = [(n + 2 * pp - kk) // ss + 1 for n, pp, kk, ss in zip((D, H, W), p, k, s_)]
        dy = rng.standard_normal((N, cout, Do, Ho, Wo)).astype(np.float32)
        xt, yt, dyt, dxt = t_from_ncdhw(x), t_empty(N, cout, Do, Ho, Wo), t_from_ncdhw(dy), t_empty(N, cin, D, H, W)
        cd = _desc(k, s_, p)
        taps = int(np.prod(k))
        mkw = lambda sc: (rng.standard_normal((cout, cin) + k) * sc / np.sqrt(cin * taps)).astype(np.float32)
        b = np.zeros(cout, np.float32)
        bp = vec(b)
        count = cout * cin * taps
        tol = 8e-6 * np.sqrt(cin * taps / 1000 + 1)

        def check(wp, w, what):
            d.call("msk_conv3d_fwd", cd, xt.msk(), vp(wp), vp(bp), yt.msk())
            e = rel_err(t_to_ncdhw(yt), O.conv3d(f8(x), f8(w), f8(b), s_, p))
            assert e < tol, (what, "fwd", e)
            d.call("msk_conv3d_dgrad", cd, dyt.msk(), vp(wp), dxt.msk(), 0)
            e = rel_err(t_to_ncdhw(dxt), O.conv3d_dgrad(f8(dy), f8(w), x.shape, s_, p))
            assert e < tol * np.sqrt(max(1.0, cout / cin)), (what, "dgrad", e)

        w1 = mkw(1.0)
        wp = vec(w1.ravel())
        check(wp, w1, "first use")
        d.prof_reset()
        d.prof_enable(True)
        check(wp, w1, "cached")
        d.prof_enable(False)
        assert not [t for t in d.prof_report() if t.startswith("pack_weights")], d.prof_report()   # no pack launch of any kind on a cached row
        w2 = mkw(37.0)
        d.h2d(wp, w2.ravel())
        check(wp, w2, "after h2d")
        w3 = mkw(1e-3)
        src = vec(w3.ravel())
        d.d2d(wp, src, count * 4)
        check(wp, w3, "after d2d")
        d.memset(wp, 0, count * 4)
        check(wp, np.zeros_like(w1), "after memset")
        d.h2d(wp, w1.ravel())
        g = mkw(1.0)
        gp, vel = vec(g.ravel()), vec(np.zeros(count, np.float32))
        check(wp, w1, "before sgd")
        d.prof_reset()
        d.prof_enable(True)
        d.call("msk_sgd_momentum", vp(wp), vp(gp), vp(vel), C.c_size_t(count), C.c_float(0.5), C.c_float(0.0), C.c_float(0.0),
               C.c_float(1.0))
        rep = d.prof_report()
        d.prof_reset()
        check(wp, w1 - 0.5 * g, "after sgd")
        d.prof_enable(False)
        assert "pack_weights_batch" in rep, rep                              # both directions rebuilt by the optimizer call ...
        assert not [t for t in d.prof_report() if t.startswith("pack_weights")], d.prof_report()   # ... none by the convolutions after it
        # eager slice update (msk_sgd_momentum_eager + _finish): the slice's rows are rebuilt on the optimizer's stream
        d.call("msk_sgd_momentum_eager", vp(wp), vp(gp), vp(vel), C.c_size_t(count), C.c_float(0.25), C.c_float(0.0), C.c_float(0.0),
               C.c_float(1.0))
        d.call("msk_sgd_momentum_finish")
        w_e = d.d2h(wp, (cout, cin) + k, np.float32)
        assert np.abs(w_e - (w1 - 0.5 * g)).max() > 1e-3
        check(wp, w_e, "after the eager update")
        m1, m2 = vec(np.zeros(count, np.float32)), vec(np.zeros(count, np.float32))
        d.call("msk_adam", vp(wp), vp(gp), vp(m1), vp(m2), C.c_size_t(count), C.c_float(1e-2), C.c_float(0.9), C.c_float(0.999),
               C.c_float(1e-8), C.c_double(0.9), C.c_double(0.999), C.c_float(0.0), C.c_float(1.0))
        w_adam = d.d2h(wp, (cout, cin) + k, np.float32)
        check(wp, w_adam, "after adam")
        d.free(wp)
        w5 = mkw(5.0)
        wp2 = vec(w5.ravel())
        check(wp2, w5, "after free + malloc")
        d.set_option("small_pack_cache", 0)
        try:
            check(wp2, w5, "cache off")
        finally:
            d.set_option("small_pack_cache", 1)


@pytest.mark.parametrize("cin,cout,src", [(64, 16, (2, 8, 8, 8)), (32, 8, (1, 5, 6, 7)), (64, 16, (1, 16, 16, 16)),
                                          (64, 16, (1, 2, 4, 32)), (32, 8, (2, 2, 3, 64)), (96, 16, (1, 2, 2, 32))])   # the last three: the LDS-staged form (W % 32 == 0)
def test_convT_bwd_bnact_equals_the_three_call_form(cin, cout, src):
    """msk_convT3d_bwd_bnact (round 5): backward of an up-convolution unit convT -> BatchNorm -> PReLU (vnet.py:133-150) with dy
    evaluated inside the data gradient's loads, against msk_affine_act_bwd_apply + msk_convT3d_dgrad + msk_convT3d_wgrad.  dout is
    a channel slice of a wider buffer (the concat gradient), dx is accumulated into."""
    from medicalseg_amd._lib import NULL_TENSOR
    d = dev()
    N, D, H, W = src
    rng = np.random.default_rng(cin + cout)
    k = s_ = (2, 2, 2)
    cd = _desc(k, s_, (0, 0, 0))
    x = rng.standard_normal((N, cin, D, H, W)).astype(np.float32)
    y = (rng.standard_normal((N, cout, 2 * D, 2 * H, 2 * W)) * 2 + 0.5).astype(np.float32)
    dwide = rng.standard_normal((N, 2 * cout, 2 * D, 2 * H, 2 * W)).astype(np.float32)
    w = (rng.standard_normal((cin, cout) + k) / np.sqrt(cin)).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    alpha = rng.uniform(0.05, 0.4, cout).astype(np.float32)
    mean, invstd = rng.standard_normal(cout).astype(np.float32), rng.uniform(0.5, 2.0, cout).astype(np.float32)
    M = N * 8 * D * H * W
    sums = (rng.standard_normal(2 * cout) * np.sqrt(M)).astype(np.float32)
    xt, yt, wide = t_from_ncdhw(x), t_from_ncdhw(y), t_from_ncdhw(dwide)
    dt = wide.channel_slice(0, cout)
    wp = vec(w.ravel())
    psc, psf, pal, pmu, pis, psm = vec(scale), vec(shift), vec(alpha), vec(mean), vec(invstd), vec(sums)
    dx0 = rng.standard_normal(x.shape).astype(np.float32)
    # three calls
    dy1 = t_empty(N, cout, 2 * D, 2 * H, 2 * W)
    dx1, dw1 = t_from_ncdhw(dx0), vec(np.full(w.size, 0.25, np.float32))
    d.call("msk_affine_act_bwd_apply", yt.msk(), vp(psc), vp(psf), NULL_TENSOR, vp(pal), vp(pmu), vp(pis), None, dt.msk(),
           vp(psm), C.c_double(M), 1, dy1.msk(), NULL_TENSOR, 0)
    d.call("msk_convT3d_wgrad", cd, xt.msk(), dy1.msk(), vp(dw1), None, 1)
    d.call("msk_convT3d_dgrad", cd, dy1.msk(), vp(wp), dx1.msk(), 1)
    # one call
    dy2 = t_empty(N, cout, 2 * D, 2 * H, 2 * W, fill=7.0)
    dx2, dw2 = t_from_ncdhw(dx0), vec(np.full(w.size, 0.25, np.float32))
    rc = d.lib.msk_convT3d_bwd_bnact(d.ctx, cd, xt.msk(), vp(wp), yt.msk(), vp(psc), vp(psf), vp(pal), vp(pmu), vp(pis), dt.msk(),
                                     vp(psm), C.c_double(M), dy2.msk(), dx2.msk(), 1, vp(dw2), 1)
    assert rc == 0, rc
    d.sync()
    assert np.array_equal(t_to_ncdhw(dy2), t_to_ncdhw(dy1))                       # the same pass, on the other stream
    assert np.array_equal(vec_back(dw2, w.size), vec_back(dw1, w.size))           # the same weight-gradient kernels on the same dy
    ref = t_to_ncdhw(dx1)
    assert rel_err(t_to_ncdhw(dx2), ref) < 2e-6, rel_err(t_to_ncdhw(dx2), ref)   # dy re-evaluated in registers: fp32 contraction only
    # float64 oracle of the whole chain
    f8 = lambda a: a.astype(np.float64)
    sh = (1, cout, 1, 1, 1)
    u = f8(y) * f8(scale).reshape(sh) + f8(shift).reshape(sh)
    du = f8(dwide[:, :cout]) * np.where(u > 0, 1.0, f8(alpha).reshape(sh))
    xh = (f8(y) - f8(mean).reshape(sh)) * f8(invstd).reshape(sh)
    dy = f8(scale).reshape(sh) * (du - f8(sums[:cout]).reshape(sh) / M - xh * f8(sums[cout:]).reshape(sh) / M)
    dx_ref = O.conv3d(dy, np.transpose(f8(w), (0, 1, 2, 3, 4)), None, s_, 0) + f8(dx0)   # convT^T = a k == s convolution with w[ci][co]
    assert rel_err(t_to_ncdhw(dx2), dx_ref) < 2e-5


@pytest.mark.parametrize("cin,cout,src", [(64, 16, (1, 16, 32, 32)), (128, 32, (1, 16, 32, 32)), (32, 16, (1, 17, 31, 33)),
                                          (256, 64, (1, 26, 26, 26)), (64, 16, (2, 4, 4, 4))])
def test_convT_fwd_ex_statistics_in_the_store_pass(cin, cout, src):
    """msk_convT3d_fwd_ex (round 5): the up-convolution's output and its BatchNorm statistics (+ finalisation) from ONE call -- in
    convT_scatter_lds_k's store pass for >= 16384 source voxels (1, 2 and 4 row tiles per wavefront, a ragged last workgroup), by
    the separate pass otherwise -- against msk_convT3d_fwd + msk_bn_stats and the float64 moments of the output."""
    from medicalseg_amd._lib import MskBnFin
    d = dev()
    N, D, H, W = src
    rng = np.random.default_rng(cin)
    k = s_ = (2, 2, 2)
    cd = _desc(k, s_, (0, 0, 0))
    x = (rng.standard_normal((N, cin, D, H, W)) + 0.3).astype(np.float32)
    w = (rng.standard_normal((cin, cout) + k) / np.sqrt(cin)).astype(np.float32)
    b = (rng.standard_normal(cout) * 3).astype(np.float32)      # a large mean per channel: the shifted sums must not cancel
    xt, wp, bp = t_from_ncdhw(x), vec(w.ravel()), vec(b)
    y1, y2 = t_empty(N, cout, 2 * D, 2 * H, 2 * W), t_empty(N, cout, 2 * D, 2 * H, 2 * W)
    st1, st2 = vec(np.zeros(2 * cout)), vec(np.zeros(2 * cout))
    d.call("msk_convT3d_fwd", cd, xt.msk(), vp(wp), vp(bp), y1.msk())
    d.call("msk_bn_stats", y1.msk(), vp(st1))
    gamma, beta = rng.uniform(0.5, 1.5, cout).astype(np.float32), rng.standard_normal(cout).astype(np.float32)
    rm, rv = vec(np.zeros(cout)), vec(np.ones(cout))
    sm, si, sc, sh = vec(np.zeros(cout)), vec(np.zeros(cout)), vec(np.zeros(cout)), vec(np.zeros(cout))
    M = N * 8 * D * H * W
    fin = MskBnFin(vec(gamma), vec(beta), 1e-5, 0.9, float(M), rm, rv, sm, si, sc, sh)
    d.call("msk_convT3d_fwd_ex", cd, xt.msk(), vp(wp), vp(bp), y2.msk(), vp(st2), C.byref(fin))
    ya, yb = t_to_ncdhw(y1), t_to_ncdhw(y2)
    assert np.array_equal(ya, yb)
    y64 = ya.astype(np.float64)
    mean, var = y64.mean(axis=(0, 2, 3, 4)), y64.var(axis=(0, 2, 3, 4))
    a, b2 = vec_back(st1, 2 * cout), vec_back(st2, 2 * cout)
    assert np.abs(b2[:cout] - mean).max() < 2e-6 * max(1.0, np.abs(mean).max()), np.abs(b2[:cout] - mean).max()
    assert rel_err(b2[cout:] / M, var) < 1e-5 and rel_err(a[cout:] / M, var) < 1e-5
    assert rel_err(vec_back(si, cout), 1.0 / np.sqrt(var + 1e-5)) < 1e-5
    assert rel_err(vec_back(sc, cout), gamma / np.sqrt(var + 1e-5)) < 1e-5
    assert rel_err(vec_back(rm, cout), 0.1 * mean) < 1e-5


def test_affine_act_fwd_amax2_second_output():
    """msk_affine_act_fwd_amax2 (round 5): BatchNorm apply + tiled residual + PReLU written into a channel slice of a wider buffer
    (the skip half of a concat) AND as a dense copy, both bitwise the one-output kernel's result; the maximum folded once."""
    d = dev()
    rng = np.random.default_rng(4)
    N, C_, D, H, W = 2, 16, 5, 6, 9
    x = rng.standard_normal((N, C_, D, H, W)).astype(np.float32)
    r = rng.standard_normal((N, 1, D, H, W)).astype(np.float32)
    scale, shift = rng.uniform(0.5, 1.5, C_).astype(np.float32), rng.standard_normal(C_).astype(np.float32)
    alpha = rng.uniform(0.1, 0.4, C_).astype(np.float32)
    xt, rt = t_from_ncdhw(x), t_from_ncdhw(r)
    ref = t_empty(N, C_, D, H, W)
    d.call("msk_affine_act_fwd", xt.msk(), vp(vec(scale)), vp(vec(shift)), rt.msk(), vp(vec(alpha)), ref.msk())
    wide, dense = t_empty(N, 2 * C_, D, H, W, fill=0.0), t_empty(N, C_, D, H, W, fill=5.0)
    amax = d.amax_new()
    d.call("msk_affine_act_fwd_amax2", xt.msk(), vp(vec(scale)), vp(vec(shift)), rt.msk(), vp(vec(alpha)),
           wide.channel_slice(C_, 2 * C_).msk(), C.c_void_p(amax), dense.msk())
    want = t_to_ncdhw(ref)
    assert np.array_equal(t_to_ncdhw(dense), want)
    got = wide.numpy()
    assert np.array_equal(got[:, C_:], want) and np.all(got[:, :C_] == 0)
    assert abs(d.d2h(amax, (64,), np.float32).max() - np.abs(want).max()) < 1e-6
