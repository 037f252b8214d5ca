"""Host-logic dry run WITHOUT a GPU: loads tests/fake_msegk.c (a no-compute stand-in built
with gcc) in place of libmsegk.so and drives the real Python stack through a full
train/eval iteration -- catching ctypes signature mismatches, shape bookkeeping errors,
arena misuse and control-flow bugs before any GPU time is spent.  Numbers are garbage by
construction; nothing numeric is asserted here."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fake_pkg(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake") / "libfake_msegk.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-o", so, os.path.join(HERE, "fake_msegk.c")])
    for m in [k for k in sys.modules if k.startswith("medicalseg_amd")]:
        del sys.modules[m]
    import importlib
    lib = importlib.import_module("medicalseg_amd._lib")
    real = lib.LIB_PATH
    lib.LIB_PATH = so
    lib._lib = None
    import medicalseg_amd
    from medicalseg_amd.device import Device
    Device._current = None
    yield medicalseg_amd
    lib.LIB_PATH = real
    lib._lib = None
    Device._current = None
    for m in [k for k in sys.modules if k.startswith("medicalseg_amd")]:
        del sys.modules[m]


def test_full_step_control_flow(fake_pkg, tmp_path):
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.core import evaluate, train
    from medicalseg_amd.datasets import SyntheticCT
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    model = VNet(num_classes=3)
    assert len(model.parameters()) == 130
    assert sum(p.size for p in model.parameters()) == 45607944
    assert len(model.state_dict()) == 178
    sched = optim.lr.PolynomialDecay(1e-3, decay_steps=10, end_lr=0, power=0.9)
    opt = optim.Momentum(sched, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    ds = SyntheticCT(num_samples=4, shape=(16, 16, 16), num_classes=3)
    val = SyntheticCT(num_samples=2, shape=(16, 16, 16), num_classes=3, mode="val")
    train(model, ds, val_dataset=val, optimizer=opt, save_dir=str(tmp_path / "out"), iters=3, batch_size=2,
          save_interval=2, log_iters=1, losses=losses, keep_checkpoint_max=1)
    assert os.path.exists(tmp_path / "out" / "iter_3" / "model.pdparams")
    assert not os.path.exists(tmp_path / "out" / "iter_2")          # rotated away (keep_checkpoint_max=1)
    assert os.path.exists(tmp_path / "out" / "best_model" / "model.pdparams")
    res = evaluate(model, val, losses, print_detail=False)
    assert "mdice" in res
    # resume parses the iteration from the directory suffix (utils.py:115-135)
    from medicalseg_amd.utils import resume
    assert resume(model, opt, str(tmp_path / "out" / "iter_3")) == 3


def test_adam_and_dice_options_control_flow(fake_pkg, tmp_path):
    """`optimizer: {type: adam}` (cvlibs/config.py:214-216) and DiceLoss(sigmoid_norm=False, weight=...)
    (dice_loss.py:36-43) drive a full iteration, checkpoint and resume through the real host stack."""
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.core import train
    from medicalseg_amd.datasets import SyntheticCT
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.utils import resume
    model = VNet(num_classes=3)
    opt = optim.Adam(1e-3, parameters=model.parameters(), weight_decay=1e-4)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss(sigmoid_norm=False, weight=[1.0, 2.0, 0.5])], [1, 1])],
              "coef": [1]}
    ds = SyntheticCT(num_samples=2, shape=(16, 16, 16), num_classes=3)
    train(model, ds, optimizer=opt, save_dir=str(tmp_path / "o"), iters=2, batch_size=1, save_interval=2, log_iters=1,
          losses=losses)
    assert abs(opt.beta1_pow - 0.9 ** 3) < 1e-12 and abs(opt.beta2_pow - 0.999 ** 3) < 1e-12
    sd = opt.state_dict()
    assert "in_tr.conv1.weight_moment1_0" in sd and "in_tr.conv1.weight_beta2_pow_acc_0" in sd
    opt2 = optim.Adam(1e-3, parameters=model.parameters())
    assert resume(model, opt2, str(tmp_path / "o" / "iter_2")) == 2
    assert abs(opt2.beta1_pow - 0.9 ** 3) < 1e-12 and not opt2.last_load["missing"]
    with pytest.raises(ValueError):
        DiceLoss(weight=[1.0, 2.0])(_logits(model), _labels())  # two weights for three classes


def _logits(model):
    from medicalseg_amd.device import to_tensor
    return model(to_tensor(np.zeros((1, 1, 16, 16, 16), np.float32)))[0]


def _labels():
    from medicalseg_amd.device import to_tensor
    return to_tensor(np.zeros((1, 16, 16, 16), np.int32))


def test_elu_control_flow(fake_pkg):
    """VNet(elu=True) (vnet.py:25-29): ELU units drive a forward/backward through the real host stack; no PReLU tensors."""
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.utils import loss_computation
    model = VNet(elu=True, num_classes=3)
    assert not [k for k in model.state_dict() if "relu" in k]
    assert len(model.parameters()) == 130 - 32          # 32 PReLU slope tensors fewer than the elu=False net
    model.train()
    logits = model(to_tensor(np.zeros((1, 1, 16, 16, 16), np.float32)))
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    ll, _ = loss_computation(logits, to_tensor(np.zeros((1, 16, 16, 16), np.int32)), losses)
    sum(ll).backward()
    model.eval()
    from medicalseg_amd.core import infer
    pred, _ = infer.inference(model, to_tensor(np.zeros((1, 1, 16, 16, 16), np.float32)))
    assert tuple(pred.shape) == (1, 1, 16, 16, 16)


def test_mri_config_shapes(fake_pkg):
    """Anisotropic MRI kernels: spatial chain 512x512x12 -> ... (vnet.py:258-265 comments),
    checked here at 1/8 scale in-plane."""
    from medicalseg_amd.models import VNet
    K = [[2, 2, 4], [2, 2, 2], [2, 2, 2], [2, 2, 2]]
    S = [[2, 2, 1], [2, 2, 1], [2, 2, 2], [2, 2, 2]]
    model = VNet(num_classes=20, kernel_size=K, stride_size=S)
    assert sum(p.size for p in model.parameters()) == 45688708
    out = model(np.zeros((1, 1, 64, 64, 12), np.float32))[0]
    assert out.shape == (1, 20, 64, 64, 12)
    acts = [a.shape for a in model._acts]
    assert acts == [(1, 16, 64, 64, 12), (1, 32, 32, 32, 9), (1, 64, 16, 16, 8), (1, 128, 8, 8, 4), (1, 256, 4, 4, 2)]
    with pytest.raises(ValueError):
        model(np.zeros((1, 2, 64, 64, 12), np.float32))


def test_deepsup_control_flow_and_config(fake_pkg, tmp_path):
    """VNetDeepSup (vnet_deepsup.py:178-281): four outputs at input size, state-dict keys in the
    reference's attribute order, out_tr_all kept out of the optimizer, and the shipped config
    (one loss type x four coefs) trains through core.train."""
    from medicalseg_amd import optimizer as optim
    from medicalseg_amd.core import train
    from medicalseg_amd.cvlibs import Config
    from medicalseg_amd.models import VNetDeepSup
    from oracle import vnet_numpy as O
    cfg = Config(os.path.join(os.path.dirname(HERE), "configs", "synthetic", "vnetdeepsup_synthetic_mri_512_512_12.yml"))
    assert cfg.dic["model"]["type"] == "VNetDeepSup"
    losses = cfg.loss
    assert len(losses["types"]) == 4 and losses["coef"] == [0.25] * 4
    assert len({id(t) for t in losses["types"]}) == 4          # separate instances -> separate CE class weights
    K, S = cfg.dic["model"]["kernel_size"], cfg.dic["model"]["stride_size"]
    model = VNetDeepSup(num_classes=20, kernel_size=K, stride_size=S)
    assert list(model.state_dict().keys()) == [n for n, _, _ in O.param_specs_deepsup(1, 20, K, S)]
    frozen = [p for p in model.parameters() if getattr(p, "frozen", False)]
    assert len(frozen) == 7 and all(p.name.startswith("out_tr_all.") for p in frozen)
    assert len(model.parameters()) == 130 + 6 + 7
    opt = optim.Momentum(1e-3, parameters=model.parameters(), momentum=0.9, weight_decay=1e-4)
    assert len(opt._parameter_list) == 136 and opt.arena is model.arena
    outs = model(np.zeros((1, 1, 64, 64, 12), np.float32))
    assert [o.shape for o in outs] == [(1, 20, 64, 64, 12)] * 4
    assert [h.shape for h in model._heads] == [(1, 20, 8, 8, 4), (1, 20, 16, 16, 8), (1, 20, 32, 32, 9)]
    from medicalseg_amd.datasets import SyntheticCT
    ds = SyntheticCT(num_samples=2, shape=(32, 32, 12), num_classes=20)
    train(model, ds, optimizer=opt, save_dir=str(tmp_path / "o"), iters=2, batch_size=1, save_interval=2, log_iters=1,
          losses=losses)
    assert os.path.exists(tmp_path / "o" / "iter_2" / "model.pdparams")
    import pickle
    sd = pickle.load(open(tmp_path / "o" / "iter_2" / "model.pdparams", "rb"))
    assert "out_tr_all.conv1.weight" in sd and "out_tr256.weight" in sd
    od = pickle.load(open(tmp_path / "o" / "iter_2" / "model.pdopt", "rb"))
    assert "out_tr64.weight_velocity_0" in od and not any(k.startswith("out_tr_all") for k in od)


def test_gradient_buckets_partition_the_arena(fake_pkg):
    """parallel.DataParallel(overlap=True): the slices handed to msk_dp_allreduce_async while backward runs are
    disjoint, each a tail of what was left, big enough, and together exactly the gradient arena."""
    from medicalseg_amd import parallel
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, VNet, VNetDeepSup
    from medicalseg_amd.utils import loss_computation
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 1, 16, 16, 16)).astype(np.float32)
    y = rng.integers(0, 3, (1, 16, 16, 16)).astype(np.int32)
    from medicalseg_amd.device import get_device
    get_device().set_option("dp_mode", 2)     # buckets need a communication stream (dp_mode 0 falls back to one all-reduce)
    for cls, nout, min_buckets in ((VNet, 1, 4), (VNetDeepSup, 4, 3)):
        model = cls(num_classes=3)
        ddp = parallel.DataParallel(model, force=True, overlap=True, bucket_bytes=16 << 20)
        losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])] * nout, "coef": [1.0 / nout] * nout}
        for _ in range(2):                                  # the bookkeeping resets between steps
            ll, _ = loss_computation(ddp(x), to_tensor(y), losses)
            sum(ll).backward()
            sent = ddp.buckets_last_step
            assert len(sent) >= min_buckets, sent
            end = model.arena.count
            for off, cnt in sent:                           # sent from the back of the arena to the front
                assert cnt > 0 and off + cnt == end, (off, cnt, end)
                end = off
            assert end == 0
            assert all(cnt * 4 >= 16 << 20 for _, cnt in sent[:-1])
            model.clear_gradients()
    # overlap off: one all-reduce of the whole arena after backward
    model = VNet(num_classes=3)
    ddp = parallel.DataParallel(model, force=True, overlap=False)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    ll, _ = loss_computation(ddp(x), to_tensor(y), losses)
    sum(ll).backward()
    assert ddp.buckets_last_step == [(0, model.arena.count)]


def test_unet3d_control_flow_and_config(fake_pkg):
    """Builder-defined UNet3D: YAML -> Config -> model, shapes, one step of the data-parallel bucket hooks."""
    from medicalseg_amd import parallel
    from medicalseg_amd.cvlibs import Config
    from medicalseg_amd.device import to_tensor
    from medicalseg_amd.utils import loss_computation
    cfg = Config(os.path.join(HERE, "..", "configs", "synthetic", "unet3d_synthetic_liver_192_192_64.yml"))
    assert cfg.dic["model"]["type"] == "UNet3D" and cfg.batch_size == 2
    built = cfg.model                       # through ComponentFactory with the keys inherited from the VNet base config
    assert type(built).__name__ == "UNet3D" and built.depth == 4
    from medicalseg_amd.models import CrossEntropyLoss, DiceLoss, MixedLoss, UNet3D
    model = UNet3D(num_classes=3, base_channels=8, depth=3)
    names = [n for n, _ in model.named_parameters()]
    assert "enc0.norm1.scale" in names and "up0.up_conv.weight" in names and "head.bias" in names
    from medicalseg_amd.device import get_device
    get_device().set_option("dp_mode", 2)
    ddp = parallel.DataParallel(model, force=True, overlap=True, bucket_bytes=1 << 10)
    x = np.zeros((2, 1, 16, 16, 8), np.float32)
    out = ddp(x)[0]
    assert out.shape == (2, 3, 16, 16, 8)
    losses = {"types": [MixedLoss([CrossEntropyLoss(), DiceLoss()], [1, 1])], "coef": [1]}
    ll, _ = loss_computation([out], to_tensor(np.zeros((2, 16, 16, 8), np.int32)), losses)
    sum(ll).backward()
    end = model.arena.count
    for off, cnt in ddp.buckets_last_step:
        assert off + cnt == end
        end = off
    assert end == 0 and len(ddp.buckets_last_step) >= 2
    with pytest.raises(ValueError):
        model(np.zeros((1, 1, 10, 16, 8), np.float32))      # 10 is not a multiple of 2^(depth-1)


def test_stale_activation_is_detected(fake_pkg):
    from medicalseg_amd._lib import MskError
    from medicalseg_amd.models import VNet
    model = VNet(num_classes=3)
    a = model(np.zeros((1, 1, 16, 16, 16), np.float32))[0]
    model(np.zeros((1, 1, 16, 16, 16), np.float32))
    with pytest.raises(MskError):
        a.numpy()


def test_preprocess_wrappers(fake_pkg):
    from medicalseg_amd import preprocess as pp
    out, sp = pp.resample(np.zeros((8, 9, 10), np.float32), spacing=[1, 2, 3], new_shape=[4, 4, 4], order=1)
    assert out.shape == (4, 4, 4) and out.dtype == np.float32 and np.allclose(sp, [2, 4.5, 7.5])
    lab, _ = pp.resample(np.zeros((8, 9, 10), np.int64), new_shape=[4, 4, 4], order=0)
    assert lab.dtype == np.int64
    assert pp.HUnorm(np.zeros((2, 3, 4))).shape == (2, 3, 4)
    assert pp.max_normalize(np.ones((2, 3, 4))).shape == (1, 2, 3, 4)
    assert pp.label_remap(np.zeros((2, 3, 4), np.int32), {1: 2}).shape == (2, 3, 4)


def test_train_profiler_options_and_step_protocol(tmp_path):
    """--profiler_options (reference utils/train_profiler.py:26-112): the option grammar, profile on at batch_range[0],
    report + optional exit at batch_range[1]."""
    from medicalseg_amd.utils import train_profiler as TP
    o = TP.ProfilerOptions("batch_range=[3, 5]; profile_path=%s; exit_on_finished=False; sorted_key=calls" % (tmp_path / "p.tsv"))
    assert o["batch_range"] == [3, 5] and o["exit_on_finished"] is False and o["sorted_key"] == "calls"
    assert TP.ProfilerOptions("batch_range=[5,3]")["batch_range"] == [10, 20]       # invalid range keeps the default
    with pytest.raises(ValueError):
        o["nope"]

    class Dev:
        def __init__(self):
            self.log = []

        def set_option(self, k, v):
            self.log.append(("opt", k, v))

        def prof_reset(self):
            self.log.append("reset")

        def prof_enable(self, on):
            self.log.append(("enable", bool(on)))

        def sync(self):
            self.log.append("sync")

        def prof_report(self):
            return {"kernel_a": (4, 2.0), "kernel_b": (9, 1.0)}

    TP.reset()
    d = Dev()
    s = "batch_range=[2, 4]; profile_path=%s; exit_on_finished=False; sorted_key=calls" % (tmp_path / "p.tsv")
    for _ in range(6):
        TP.add_profiler_step(s, d)
    assert ("enable", True) in d.log and ("enable", False) in d.log
    assert d.log.index(("enable", True)) < d.log.index("sync") < d.log.index(("enable", False))
    lines = [l for l in open(tmp_path / "p.tsv").read().splitlines() if not l.startswith("#")]
    assert lines[0].startswith("kernel_b\t9") and lines[1].startswith("kernel_a\t4")
    TP.reset()
    with pytest.raises(SystemExit):
        for _ in range(6):
            TP.add_profiler_step("batch_range=[1, 2]; profile_path=%s" % (tmp_path / "q.tsv"), d)
    TP.reset()
    TP.add_profiler_step(None, d)      # disabled
