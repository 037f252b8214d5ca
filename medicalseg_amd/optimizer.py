"""Optimizers and LR schedules with paddle.optimizer's surface as used by the reference
(cvlibs/config.py:156-232, core/train.py:140-151, utils/utils.py:125)."""
import ctypes as C
import os
import math

import numpy as np

__all__ = ["Momentum", "SGD", "Adam", "lr"]
OPTIMIZERS = ("Momentum", "SGD", "Adam")     # names accepted as an optimizer `type` (cvlibs/config.py); `lr` is the scheduler namespace


def _warn_unused(cls_name, kw):
    """paddle.optimizer.* options this package does not implement (grad_clip, lazy_mode, multi_precision, use_nesterov ...):
    the reference passes them through to Paddle (cvlibs/config.py:217-224); dropping one silently would train differently
    without a trace."""
    if kw:
        import warnings
        warnings.warn("%s: option(s) %s are not implemented by medicalseg_amd and are IGNORED" % (cls_name, sorted(kw)))



class _LR:
    """namespace mirroring paddle.optimizer.lr"""

    class LRScheduler:
        def __init__(self, learning_rate=0.1, last_epoch=-1):
            self.base_lr = float(learning_rate)
            self.last_epoch = last_epoch
            self.last_lr = self.base_lr
            self.step()

        def __call__(self):
            return self.last_lr

        def get_lr(self):
            raise NotImplementedError

        def step(self, epoch=None):
            self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
            self.last_lr = self.get_lr()

        def state_dict(self):
            return {"last_epoch": self.last_epoch, "last_lr": self.last_lr}

        def set_state_dict(self, sd):
            self.last_epoch = sd.get("last_epoch", self.last_epoch)
            self.last_lr = sd.get("last_lr", self.get_lr())

    class PolynomialDecay(LRScheduler):
        """lr = (lr0 - end)*(1 - min(t,T)/T)^power + end; t advances on every step()
        (core/train.py:150-151); first optimizer step uses lr0 (App. B.8 vi)."""

        def __init__(self, learning_rate, decay_steps, end_lr=0.0001, power=1.0, cycle=False, last_epoch=-1):
            self.decay_steps, self.end_lr, self.power, self.cycle = decay_steps, end_lr, power, cycle
            super().__init__(learning_rate, last_epoch)

        def get_lr(self):
            t, T = self.last_epoch, self.decay_steps
            if self.cycle:
                div = math.ceil(t / float(T)) if t > 0 else 1
                T = T * div
            else:
                t = min(t, T)
            return (self.base_lr - self.end_lr) * ((1 - float(t) / float(T)) ** self.power) + self.end_lr

    class PiecewiseDecay(LRScheduler):
        def __init__(self, boundaries, values, last_epoch=-1):
            self.boundaries, self.values = boundaries, values
            super().__init__(values[0], last_epoch)

        def get_lr(self):
            for i, b in enumerate(self.boundaries):
                if self.last_epoch < b:
                    return self.values[i]
            return self.values[len(self.values) - 1]

    class StepDecay(LRScheduler):
        def __init__(self, learning_rate, step_size, gamma=0.1, last_epoch=-1):
            self.step_size, self.gamma = step_size, gamma
            super().__init__(learning_rate, last_epoch)

        def get_lr(self):
            return self.base_lr * (self.gamma ** (self.last_epoch // self.step_size))


lr = _LR


class Momentum:
    """paddle.optimizer.Momentum(lr, parameters, momentum, weight_decay: float = L2):
    g += wd*p; v = mu*v + g; p -= lr*v on EVERY trainable tensor (BN, PReLU, biases
    included -- App. B.8 v).  All parameters live in one flat arena, so a step is a single
    kernel over 45.6 M floats (K10)."""

    def __init__(self, learning_rate=0.001, momentum=0.9, parameters=None, weight_decay=None, **kw):
        _warn_unused(type(self).__name__, kw)
        if not parameters:
            raise ValueError("parameters must be a non-empty list")
        # tensors no forward path reaches get no gradient and are skipped by paddle's optimizer
        parameters = [p for p in parameters if not getattr(p, "frozen", False)]
        arenas = {id(p.arena): p.arena for p in parameters}
        if len(arenas) != 1 or None in [p.arena for p in parameters]:
            raise ValueError("all parameters must belong to one built model (one ParamArena)")
        self.arena = next(iter(arenas.values()))
        self._learning_rate = learning_rate
        self.momentum = float(momentum)
        self.weight_decay = float(weight_decay) if weight_decay else 0.0
        self._parameter_list = list(parameters)
        dev = self.arena.dev
        self.velocity_ptr = dev.malloc(max(self.arena.count, 4) * 4)
        dev.memset(self.velocity_ptr, 0, max(self.arena.count, 4) * 4)

        self._eager = False
        self.eager_min_floats = int(os.environ.get("MSEGK_EAGER_MIN_FLOATS", 1 << 20))   # blocks below this stay with step() (launch count; A/B: 0 = every block)
        self._eager_done = []     # [(lo, hi)] slices of the arena already updated during this backward pass

    def get_lr(self):
        if isinstance(self._learning_rate, _LR.LRScheduler):
            return self._learning_rate()
        return float(self._learning_rate)

    def set_lr(self, value):
        self._learning_rate = float(value)

    # -- eager mode (round 5) ---------------------------------------------------------------
    def enable_eager(self, model, on=True):
        """Opt-in: update a block's parameters (and re-pack its convolution weights) on the weight-gradient stream as soon as
        that block's backward has been enqueued (`model._grad_ready_hooks`, the hook data parallelism uses for its buckets),
        instead of one pass over the arena at `step()` -- which then only joins.  The reference's loop is loss.backward();
        optimizer.step() (core/train.py:139-140): nothing reads a block's weights between its data gradient and the next
        forward, so the results are bitwise those of the plain order (tests/test_gpu_model.py); what changes is WHEN the
        parameters change -- during backward -- hence a caller that inspects parameters between backward() and step(), or
        calls backward() without step() (gradient accumulation), must leave it off.  One rank only: with more ranks the
        gradients are final after the all-reduce, not after backward.  core.train() and bench.py switch it on at one rank."""
        if not on:
            self._eager = False
            return False
        if getattr(model, "arena", None) is not self.arena or not hasattr(model, "_grad_ready_hooks"):
            return False
        if self.arena.dev.world > 1:
            return False
        if not any(getattr(h, "__self__", None) is self for h in model._grad_ready_hooks):
            model._grad_ready_hooks.append(self._block_ready)
        if not self._eager:
            from .utils import logger
            logger.info("optimizer: eager mode on -- parameters of a block change DURING backward (right behind its weight "
                        "gradients); results are bitwise those of loss.backward(); optimizer.step()")
        self._eager = True
        self._slices = {}
        return True

    def _block_slice(self, block):
        """[lo, hi) floats of the arena that hold exactly this block's parameters (None when they are not one run)"""
        key = id(block)
        if key not in self._slices:
            ps = [p for p in block.parameters() if p.arena is self.arena]
            sl = None
            if ps:
                lo = min(p.offset for p in ps)
                hi = max(p.offset + ((p.size + 3) & ~3) for p in ps)
                if sum((p.size + 3) & ~3 for p in ps) == hi - lo:
                    sl = (lo, hi)
            self._slices[key] = sl
        return self._slices[key]

    def _block_ready(self, model, block):
        if not self._eager or not model.training:
            return
        sl = self._block_slice(block)
        if sl is not None and sl in self._eager_done:
            # the SAME block reported twice with no step() in between: a second backward() (gradient accumulation, a custom
            # loop) -- its parameters were already updated from the first one's gradients; silently skipping it would drop
            # the second gradient (advisor, round 5)
            raise RuntimeError("optimizer eager mode: backward() ran twice without optimizer.step() in between; eager updates "
                               "apply a block's gradient during backward -- call enable_eager(model, on=False) for gradient "
                               "accumulation or custom loops")
        if sl is None or any(lo < sl[1] and sl[0] < hi for lo, hi in self._eager_done):
            return
        if sl[1] - sl[0] < self.eager_min_floats:
            return      # small blocks (in_tr, down_tr32/64, up_tr64/32, out_tr: 7 MB of 182) wait for step(): one update per run of them
        a = self.arena
        lo, hi = sl
        a.dev.call("msk_sgd_momentum_eager", C.c_void_p(a.value_ptr + 4 * lo), C.c_void_p(a.grad_ptr + 4 * lo),
                   C.c_void_p(self.velocity_ptr + 4 * lo), C.c_size_t(hi - lo), C.c_float(self.get_lr()),
                   C.c_float(self.momentum), C.c_float(self.weight_decay), C.c_float(a.grad_scale))
        self._eager_done.append(sl)

    def step(self):
        a = self.arena
        if self._eager_done:
            # what the hooks did not cover (parameters outside the reported blocks), then the join
            done, self._eager_done = sorted(self._eager_done), []
            pos = 0
            for lo, hi in done + [(a.count, a.count)]:
                if lo > pos:
                    a.dev.call("msk_sgd_momentum_eager", C.c_void_p(a.value_ptr + 4 * pos), C.c_void_p(a.grad_ptr + 4 * pos),
                               C.c_void_p(self.velocity_ptr + 4 * pos), C.c_size_t(lo - pos), C.c_float(self.get_lr()),
                               C.c_float(self.momentum), C.c_float(self.weight_decay), C.c_float(a.grad_scale))
                pos = max(pos, hi)
            a.dev.call("msk_sgd_momentum_finish")
            return
        a.dev.call("msk_sgd_momentum", C.c_void_p(a.value_ptr), C.c_void_p(a.grad_ptr), C.c_void_p(self.velocity_ptr),
                   C.c_size_t(a.count), C.c_float(self.get_lr()), C.c_float(self.momentum),
                   C.c_float(self.weight_decay), C.c_float(a.grad_scale))

    def clear_grad(self):
        self.arena.zero_grad()

    clear_gradients = clear_grad

    def state_dict(self):
        dev = self.arena.dev
        flat = dev.d2h(self.velocity_ptr, (self.arena.count,), np.float32)
        sd = {}
        for p in self.arena.params:
            sd[p.name + "_velocity_0"] = flat[p.offset:p.offset + p.size].reshape(p.shape).copy()
        if isinstance(self._learning_rate, _LR.LRScheduler):
            sd["LR_Scheduler"] = self._learning_rate.state_dict()
        # not a Paddle key: the counter of the Dropout3D mask stream, so that a resumed run draws the masks an
        # uninterrupted one would have drawn (Paddle's own optimizer state carries no RNG state either way)
        from . import nn as _nn
        sd["@msegk_dropout_step"] = np.array([_nn.Dropout3D.step], dtype=np.int64)
        return sd

    def set_state_dict(self, sd, name_map=None):
        """Velocity keys are `<structured parameter name>_velocity_0` -- what this package's own `state_dict` writes -- or, in a
        model.pdopt written by PADDLE, `<internal parameter name>_velocity_0` (`conv3d_0.w_0_velocity_0`).  `name_map` is the
        `StructuredToParameterName@@` table of the sibling model.pdparams ({structured name: internal name}; reference
        utils/utils.py:115-135 loads both files, `utils.resume` passes the table on): with it a Paddle-written optimizer
        state resumes with its momentum.  Keys that match neither way are reported, never skipped silently."""
        dev = self.arena.dev
        flat = dev.d2h(self.velocity_ptr, (self.arena.count,), np.float32)
        used, missing = set(), []
        name_map = name_map or {}
        for p in self.arena.params:
            k = p.name + "_velocity_0"
            if k not in sd and p.name in name_map:
                k = str(name_map[p.name]) + "_velocity_0"
            if k in sd:
                v = np.asarray(sd[k], dtype=np.float32)
                if v.size != p.size:
                    raise ValueError("optimizer state %r has %d elements, parameter %s has %d" % (k, v.size, p.name, p.size))
                flat[p.offset:p.offset + p.size] = v.reshape(-1)
                used.add(k)
            else:
                missing.append(k)
        dev.h2d(self.velocity_ptr, flat)
        unexpected = [k for k in sd if k not in used and k not in ("LR_Scheduler", "@msegk_dropout_step")]
        if missing or unexpected:
            from .utils import logger
            logger.warning("optimizer state: %d velocity tensors missing (momentum restarts from zero for them), %d "
                           "unexpected keys (e.g. %s)%s" % (len(missing), len(unexpected), unexpected[:2],
                           " -- looks like a Paddle-written model.pdopt (internal parameter names)" if unexpected and
                           not used else ""))
        self.last_load = {"missing": missing, "unexpected": unexpected}
        if "@msegk_dropout_step" in sd:
            from . import nn as _nn
            _nn.Dropout3D.step = int(np.asarray(sd["@msegk_dropout_step"]).ravel()[0])
        if "LR_Scheduler" in sd and isinstance(self._learning_rate, _LR.LRScheduler):
            self._learning_rate.set_state_dict(sd["LR_Scheduler"])


class SGD(Momentum):
    def __init__(self, learning_rate=0.001, parameters=None, weight_decay=None, **kw):
        super().__init__(learning_rate, 0.0, parameters, weight_decay, **kw)


class Adam:
    """paddle.optimizer.Adam(lr, beta1=0.9, beta2=0.999, epsilon=1e-8, parameters, weight_decay: float = L2) as the
    reference's `optimizer: {type: adam}` branch builds it (cvlibs/config.py:214-216): one kernel over the flat arena,
    bias-corrected with the running powers beta^t that Paddle keeps as `*_beta{1,2}_pow_acc_0` (here two host doubles,
    identical for every tensor because all tensors step together)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, parameters=None, weight_decay=None,
                 **kw):
        _warn_unused("Adam", kw)
        if not parameters:
            raise ValueError("parameters must be a non-empty list")
        parameters = [p for p in parameters if not getattr(p, "frozen", False)]
        arenas = {id(p.arena): p.arena for p in parameters}
        if len(arenas) != 1 or None in [p.arena for p in parameters]:
            raise ValueError("all parameters must belong to one built model (one ParamArena)")
        self.arena = next(iter(arenas.values()))
        self._learning_rate = learning_rate
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.weight_decay = float(weight_decay) if weight_decay else 0.0
        self._parameter_list = list(parameters)
        self.beta1_pow, self.beta2_pow = self.beta1, self.beta2  # the powers the NEXT step uses (t = 1)
        dev = self.arena.dev
        n = max(self.arena.count, 4) * 4
        self.moment1_ptr, self.moment2_ptr = dev.malloc(n), dev.malloc(n)
        dev.memset(self.moment1_ptr, 0, n)
        dev.memset(self.moment2_ptr, 0, n)

    get_lr = Momentum.get_lr
    set_lr = Momentum.set_lr

    def step(self):
        a = self.arena
        a.dev.call("msk_adam", C.c_void_p(a.value_ptr), C.c_void_p(a.grad_ptr), C.c_void_p(self.moment1_ptr),
                   C.c_void_p(self.moment2_ptr), C.c_size_t(a.count), C.c_float(self.get_lr()), C.c_float(self.beta1),
                   C.c_float(self.beta2), C.c_float(self.epsilon), C.c_double(self.beta1_pow),
                   C.c_double(self.beta2_pow), C.c_float(self.weight_decay), C.c_float(a.grad_scale))
        self.beta1_pow *= self.beta1
        self.beta2_pow *= self.beta2

    def clear_grad(self):
        self.arena.zero_grad()

    clear_gradients = clear_grad

    def state_dict(self):
        dev = self.arena.dev
        m1 = dev.d2h(self.moment1_ptr, (self.arena.count,), np.float32)
        m2 = dev.d2h(self.moment2_ptr, (self.arena.count,), np.float32)
        sd = {}
        for p in self.arena.params:
            sd[p.name + "_moment1_0"] = m1[p.offset:p.offset + p.size].reshape(p.shape).copy()
            sd[p.name + "_moment2_0"] = m2[p.offset:p.offset + p.size].reshape(p.shape).copy()
            sd[p.name + "_beta1_pow_acc_0"] = np.array([self.beta1_pow], dtype=np.float32)
            sd[p.name + "_beta2_pow_acc_0"] = np.array([self.beta2_pow], dtype=np.float32)
        sd["@msegk_beta_pows"] = np.array([self.beta1_pow, self.beta2_pow], dtype=np.float64)
        if isinstance(self._learning_rate, _LR.LRScheduler):
            sd["LR_Scheduler"] = self._learning_rate.state_dict()
        from . import nn as _nn
        sd["@msegk_dropout_step"] = np.array([_nn.Dropout3D.step], dtype=np.int64)
        return sd

    def set_state_dict(self, sd, name_map=None):
        """name_map: `StructuredToParameterName@@` of the sibling model.pdparams -- a Paddle-written model.pdopt names its
        accumulators after the internal parameter names (see Momentum.set_state_dict)."""
        dev = self.arena.dev
        m1 = dev.d2h(self.moment1_ptr, (self.arena.count,), np.float32)
        m2 = dev.d2h(self.moment2_ptr, (self.arena.count,), np.float32)
        used, missing = set(), []
        name_map = name_map or {}
        for p in self.arena.params:
            for buf, suffix in ((m1, "_moment1_0"), (m2, "_moment2_0")):
                k = p.name + suffix
                if k not in sd and p.name in name_map:
                    k = str(name_map[p.name]) + suffix
                if k in sd:
                    buf[p.offset:p.offset + p.size] = np.asarray(sd[k], dtype=np.float32).reshape(-1)
                    used.add(k)
                else:
                    missing.append(k)
            for suffix in ("_beta1_pow_acc_0", "_beta2_pow_acc_0"):
                for nm in (p.name, str(name_map.get(p.name, ""))):
                    if nm and nm + suffix in sd:
                        used.add(nm + suffix)
        dev.h2d(self.moment1_ptr, m1)
        dev.h2d(self.moment2_ptr, m2)
        if "@msegk_beta_pows" in sd:
            self.beta1_pow, self.beta2_pow = [float(v) for v in np.asarray(sd["@msegk_beta_pows"]).ravel()[:2]]
        else:
            first = self.arena.params[0].name
            if first + "_beta1_pow_acc_0" not in sd and first in name_map:
                first = str(name_map[first])
            if first + "_beta1_pow_acc_0" in sd:
                self.beta1_pow = float(np.asarray(sd[first + "_beta1_pow_acc_0"]).ravel()[0])
                self.beta2_pow = float(np.asarray(sd[first + "_beta2_pow_acc_0"]).ravel()[0])
        unexpected = [k for k in sd if k not in used and k not in ("LR_Scheduler", "@msegk_dropout_step",
                                                                   "@msegk_beta_pows")]
        if missing or unexpected:
            from .utils import logger
            logger.warning("optimizer state: %d moment tensors missing (they restart from zero), %d unexpected keys "
                           "(e.g. %s)" % (len(missing), len(unexpected), unexpected[:2]))
        self.last_load = {"missing": missing, "unexpected": unexpected}
        if "@msegk_dropout_step" in sd:
            from . import nn as _nn
            _nn.Dropout3D.step = int(np.asarray(sd["@msegk_dropout_step"]).ravel()[0])
        if "LR_Scheduler" in sd and isinstance(self._learning_rate, _LR.LRScheduler):
            self._learning_rate.set_state_dict(sd["LR_Scheduler"])
