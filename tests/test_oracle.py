"""Pins the CPU oracle (runs without a GPU):
  * model/loss/optimizer restatement vs torch-CPU autograd (float64) -- the reference was
    ported from torch and aligned against it (vnet.py:1-3,285-294); PARITY UNPINNED by the
    reference itself (no Paddle here, no reference tests);
  * loss known-answer vector of SURVEY.md Appendix C;
  * preprocessing restatement vs goldens captured from the reference's own code.
"""
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_numpy as P
from oracle import vnet_numpy as O
from oracle.vnet_torch import TorchVNet, TorchVNetDeepSup, torch_mixed_loss

HERE = os.path.dirname(os.path.abspath(__file__))
SITES = [("down_tr128", 128), ("down_tr256", 256), ("up_tr256.x", 256), ("up_tr256.skip", 128),
         ("up_tr128.x", 256), ("up_tr128.skip", 64)]


def test_param_count_matches_reference_architecture():
    specs = O.param_specs(1, 3)
    trainable = [s for s in specs if s[2] in O.TRAINABLE_KINDS]
    assert len(trainable) == 130
    assert sum(int(np.prod(s[1])) for s in trainable) == 45607944
    assert len([s for s in specs if s[2] in ("bn_mean", "bn_var")]) == 48
    mri_k = ((2, 2, 4), (2, 2, 2), (2, 2, 2), (2, 2, 2))                # vnet_mri_spine_seg_512_512_12_15k.yml:9
    specs20 = O.param_specs(1, 20, mri_k)
    assert sum(int(np.prod(s[1])) for s in specs20 if s[2] in O.TRAINABLE_KINDS) == 45688708


@pytest.mark.parametrize("train", [True, False])
def test_oracle_vs_torch_autograd(train):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    S, ncls, N = (16, 16, 16), 3, 2
    K = Sd = ((2, 2, 2),) * 4
    rng = np.random.default_rng(0)
    params = O.init_params(1, 1, ncls, K, Sd)
    x = rng.standard_normal((N, 1) + S).astype(np.float32)
    y = rng.integers(0, ncls, (N,) + S).astype(np.int32)
    masks = {s: (rng.random((N, c)) < 0.5).astype(np.float32) * 2.0 for s, c in SITES}
    m = O.VNetOracle(params, 1, ncls, K, Sd)
    lg = m.forward(x, train=train, dropout_masks=masks)
    L = O.MixedLossOracle()
    ll, per, dz = L(lg, y)
    g = m.backward(dz)
    tm = TorchVNet(1, ncls, K, Sd).double()
    tm.load_oracle_params({k: np.asarray(v, dtype=np.float64) for k, v in params.items()})
    tm.train(train)
    tl = tm(torch.tensor(x, dtype=torch.float64), masks if train else None)
    ce, dl, tper = torch_mixed_loss(tl, torch.tensor(y), torch.tensor(L.weight))
    (ce + dl).backward()
    assert np.abs(tl.detach().numpy() - lg).max() < 1e-10
    assert abs(float(ce.detach()) - ll[0]) < 1e-10 and abs(float(dl.detach()) - ll[1]) < 1e-10
    gmax = max(np.abs(v).max() for v in g.values())
    for k, v in tm.named_oracle_grads().items():
        assert np.abs(v - g[k]).max() < 1e-9 * gmax + 1e-12 * np.abs(v).max(), k
    if train:
        sd = tm.state_dict()
        for k in m.p:
            if k.endswith("._mean"):
                assert np.abs(sd[k.replace("._mean", ".running_mean")].numpy() - m.p[k]).max() < 1e-12


def test_oracle_anisotropic_mri_config_vs_torch():
    K = ((2, 2, 4), (2, 2, 2), (2, 2, 2), (2, 2, 2))
    Sd = ((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2))
    rng = np.random.default_rng(1)
    params = O.init_params(3, 1, 5, K, Sd)
    x = rng.standard_normal((1, 1, 32, 32, 12)).astype(np.float32)
    m = O.VNetOracle(params, 1, 5, K, Sd)
    lg = m.forward(x, train=False, record=False)
    assert lg.shape == (1, 5, 32, 32, 12)
    tm = TorchVNet(1, 5, K, Sd).double()
    tm.load_oracle_params({k: np.asarray(v, dtype=np.float64) for k, v in params.items()})
    tm.eval()
    with torch.no_grad():
        tl = tm(torch.tensor(x, dtype=torch.float64))
    assert np.abs(tl.numpy() - lg).max() < 1e-10


def test_deepsup_oracle_vs_torch_autograd():
    """VNetDeepSup (vnet_deepsup.py:257-281): four outputs, loss coef 0.25 each
    (vnetdeepsup_mri_spine_seg_512_512_12_15k.yml:20), anisotropic MRI kernels."""
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    K = ((2, 2, 4), (2, 2, 2), (2, 2, 2), (2, 2, 2))
    Sd = ((2, 2, 1), (2, 2, 1), (2, 2, 2), (2, 2, 2))
    S, ncls, N = (16, 16, 12), 5, 2
    rng = np.random.default_rng(3)
    params = O.init_params_deepsup(2, 1, ncls, K, Sd)
    specs = O.param_specs_deepsup(1, ncls, K, Sd)
    # 130 trunk tensors + 3 heads x (w, b) + the 7 never-used out_tr_all tensors
    assert len([s for s in specs if s[2] in O.TRAINABLE_KINDS]) == 130 + 6 + 7
    x = rng.standard_normal((N, 1) + S).astype(np.float32)
    y = rng.integers(0, ncls, (N,) + S).astype(np.int32)
    masks = {s: (rng.random((N, c)) < 0.5).astype(np.float32) * 2.0 for s, c in SITES}
    m = O.VNetDeepSupOracle(params, 1, ncls, K, Sd)
    outs = m.forward(x, train=True, dropout_masks=masks)
    assert [o.shape for o in outs] == [(N, ncls) + S] * 4
    Ls = [O.MixedLossOracle(outer_coef=0.25) for _ in outs]
    res = [L(o, y) for L, o in zip(Ls, outs)]
    g = m.backward([r[2] for r in res])
    tm = TorchVNetDeepSup(1, ncls, K, Sd).double()
    tm.load_oracle_params({k: np.asarray(v, dtype=np.float64) for k, v in params.items()})
    tm.train(True)
    touts = tm(torch.tensor(x, dtype=torch.float64), masks)
    total = 0
    for o, to, L, r in zip(outs, touts, Ls, res):
        assert np.abs(to.detach().numpy() - o).max() < 1e-10
        ce, dl, _ = torch_mixed_loss(to, torch.tensor(y), torch.tensor(L.weight))
        assert abs(0.25 * float(ce.detach()) - r[0][0]) < 1e-10 and abs(0.25 * float(dl.detach()) - r[0][1]) < 1e-10
        total = total + 0.25 * (ce + dl)
    total.backward()
    tg = tm.named_oracle_grads()
    assert not any(k.startswith("out_tr_all") for k in tg) and not any(k.startswith("out_tr_all") for k in g)
    assert set(tg) == set(m.trainable)
    gmax = max(np.abs(v).max() for v in g.values())
    for k, v in tg.items():
        assert np.abs(v - g[k]).max() < 1e-9 * gmax + 1e-12 * np.abs(v).max(), k


def test_sgd_and_poly_lr_vs_torch():
    rng = np.random.default_rng(2)
    p0 = rng.standard_normal(50)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.SGD([tp], lr=1e-3, momentum=0.9, weight_decay=1e-4)
    params, vel = {"p": p0.copy()}, {}
    for step in range(4):
        g = rng.standard_normal(50)
        lr = O.poly_lr(step, 1e-3, 10, 0.0, 0.9)
        for grp in opt.param_groups:
            grp["lr"] = lr
        tp.grad = torch.tensor(g)
        opt.step()
        O.sgd_momentum_step(params, {"p": g}, vel, lr, 0.9, 1e-4)
        assert np.abs(tp.detach().numpy() - params["p"]).max() < 1e-14
    assert O.poly_lr(0, 1e-3, 15000) == 1e-3 and O.poly_lr(15000, 1e-3, 15000) == 0.0
    assert abs(O.poly_lr(7500, 1e-3, 15000) - 1e-3 * 0.5 ** 0.9) < 1e-18


def test_loss_known_answer_appendix_c():
    rng = np.random.default_rng(0)
    z = rng.standard_normal((1, 3, 2, 3, 4)).astype(np.float32)
    y = rng.integers(0, 3, (1, 2, 3, 4)).astype(np.int32)
    L = O.MixedLossOracle()
    ll, per, dz = L(z, y)
    assert np.allclose(L.weight, [2.47594326, 1.75639001, 1.86110799], rtol=1e-8)
    assert abs(ll[0] - 1.4261519761109072) < 1e-12 and abs(ll[1] - 0.5177017349991624) < 1e-12
    assert np.allclose(per, [0.33488489, 0.69615749, 0.41585241], atol=1e-8)
    assert np.allclose(dz[0, :, 0, 0, 0], [0.00738261, -0.0318339, 0.02601402], atol=1e-8)
    # class weights are cached from the first call
    w0 = L.weight.copy()
    L(z * 3, y)
    assert np.array_equal(L.weight, w0)


def test_preprocess_restatement_vs_reference_goldens():
    g = np.load(os.path.join(HERE, "golden", "preprocess_golden.npz"))
    for i in range(5):
        o1, sp = P.resample(g[f"rs{i}_img"], g[f"rs{i}_shape"], 1, spacing=[0.7, 0.8, 2.5])
        ref = g[f"rs{i}_o1"]
        assert o1.dtype == ref.dtype and np.abs(o1 - ref).max() <= 1e-6 * np.abs(ref).max()
        assert np.allclose(sp, g[f"rs{i}_spacing"])
        o0, _ = P.resample(g[f"rs{i}_lab"], g[f"rs{i}_shape"], 0)
        assert o0.dtype == np.int32 and np.array_equal(o0, g[f"rs{i}_o0"])
        of0, _ = P.resample(g[f"rs{i}_img"], g[f"rs{i}_shape"], 0)
        assert np.array_equal(of0, g[f"rs{i}_of0"])
    _, sp = P.resample(np.zeros((8, 8, 8), np.float32), [4, 4, 4], 1, spacing=[9.0, 1.0, 2.0, 3.0])
    assert np.allclose(sp, g["rs_sp4"])
    assert np.array_equal(P.HUnorm(g["hu_in"]), g["hu_out"])
    assert np.array_equal(g["hu_out"][:6], np.array([0, 0, 170, 255, 0, 255], np.float32))
    assert np.array_equal(P.HUnorm(g["hu_vol_in"]), g["hu_vol_out"])
    assert np.array_equal(P.HUnorm(g["hu_vol_in"], -1000, 400, -1500), g["hu_vol_out_custom"])
    assert np.array_equal(P.normalize(g["nm_in"]), g["nm_out_auto"])
    assert np.array_equal(P.normalize(g["nm_in"], 0, 2650), g["nm_out_bounds"])
    assert np.array_equal(P.label_remap(g["lr_in"], {1: 0, 2: 1, 3: 1, 5: 2}), g["lr_out"])
    assert np.array_equal(P.label_remap(g["lr_in"], {1: 2, 2: 3}), g["lr_out_chain"])
    ct, _ = P.resample(P.HUnorm(g["pipe_ct_in"]), [16, 16, 16], 1)
    assert np.abs(ct - g["pipe_ct_out"]).max() < 1e-4


def test_adam_vs_torch():
    """oracle adam_step (Paddle's kernel form, cvlibs/config.py:214-216) == torch.optim.Adam with L2 weight decay."""
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(64)
    tp = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.Adam([tp], lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    params, m1, m2 = {"p": p0.copy()}, {}, {}
    for t in range(1, 6):
        g = rng.standard_normal(64)
        tp.grad = torch.tensor(g)
        opt.step()
        O.adam_step(params, {"p": g}, m1, m2, t, 2e-3, 0.9, 0.999, 1e-8, 1e-4)
        assert np.abs(tp.detach().numpy() - params["p"]).max() < 1e-13


@pytest.mark.parametrize("sigmoid_norm,weighted", [(False, False), (True, True), (False, True)])
def test_dice_options_vs_torch_autograd(sigmoid_norm, weighted):
    """DiceLoss(sigmoid_norm=False) / DiceLoss(weight=...) (dice_loss.py:36-43,68-69) restated in the oracle against
    torch autograd of the same formula."""
    rng = np.random.default_rng(11)
    z = rng.standard_normal((2, 3, 3, 4, 5))
    y = rng.integers(0, 3, (2, 3, 4, 5))
    w = np.array([0.5, 2.0, 1.25]) if weighted else None
    loss, per, dz = O.dice(z, y, sigmoid_norm=sigmoid_norm, weight=w)
    zt = torch.tensor(z, requires_grad=True)
    p = torch.sigmoid(zt) if sigmoid_norm else torch.softmax(zt, 1)
    t = torch.nn.functional.one_hot(torch.tensor(y), 3).permute(0, 4, 1, 2, 3).double()
    pf, tf = p.transpose(0, 1).reshape(3, -1), t.transpose(0, 1).reshape(3, -1)
    inter = (pf * tf).sum(-1)
    if w is not None:
        inter = torch.tensor(w) * inter
    tper = 2 * inter / torch.clamp((pf * pf).sum(-1) + (tf * tf).sum(-1), min=1e-6)
    tl = 1 - tper.mean()
    tl.backward()
    assert abs(loss - tl.item()) < 1e-14
    assert np.abs(per - tper.detach().numpy()).max() < 1e-14
    assert np.abs(dz - zt.grad.numpy()).max() < 1e-15
    if sigmoid_norm and not weighted:
        pass
    # the default arguments still take the original code path
    l0, p0, d0 = O.dice(z, y)
    l1, p1, d1 = O._dice_general(z, y, 1e-6, True, None)
    assert abs(l0 - l1) < 1e-15 and np.abs(d0 - d1).max() < 1e-16


def test_oracle_elu_branch_vs_torch_autograd():
    """ELUCons(elu=True) (vnet.py:25-29): the oracle's ELU branch against the torch restatement with its PReLU modules
    replaced by torch.nn.ELU -- logits and every gradient."""
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    S, ncls, N = (16, 16, 16), 3, 1
    K = Sd = ((2, 2, 2),) * 4
    rng = np.random.default_rng(3)
    params = {k: v for k, v in O.init_params(2, 1, ncls, K, Sd).items() if "relu" not in k}
    x = rng.standard_normal((N, 1) + S).astype(np.float32)
    y = rng.integers(0, ncls, (N,) + S).astype(np.int32)
    m = O.VNetOracle(params, 1, ncls, K, Sd)
    m.elu = True
    m.trainable = [n for n in m.trainable if "relu" not in n]
    lg = m.forward(x, train=False)
    L = O.MixedLossOracle()
    ll, per, dz = L(lg, y)
    g = m.backward(dz)
    tm = TorchVNet(1, ncls, K, Sd).double()
    full = O.init_params(2, 1, ncls, K, Sd)
    tm.load_oracle_params({k: np.asarray(v, dtype=np.float64) for k, v in full.items()})
    for parent in list(tm.modules()):
        for name, child in list(parent.named_children()):
            if isinstance(child, torch.nn.PReLU):
                setattr(parent, name, torch.nn.ELU())
    tm.train(False)
    tl = tm(torch.tensor(x, dtype=torch.float64), None)
    ce, dl, tper = torch_mixed_loss(tl, torch.tensor(y), torch.tensor(L.weight))
    (ce + dl).backward()
    assert np.abs(tl.detach().numpy() - lg).max() < 1e-10
    gmax = max(np.abs(v).max() for v in g.values())
    tg = tm.named_oracle_grads()
    assert set(g) <= set(tg) | set(), sorted(set(g) - set(tg))[:3]
    for k, v in g.items():
        assert np.abs(tg[k] - v).max() < 1e-9 * gmax + 1e-12 * np.abs(v).max(), k
