"""Single-node data parallelism: one process per GPU, RCCL over xGMI.

Replaces paddle.distributed.fleet.init / distributed_model / distributed_optimizer
(reference core/train.py:81-85) and paddle.distributed.ParallelEnv (:69-70).

Launch contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the
environment (what ``python -m torch.distributed.run`` exports).  The RCCL unique id is
handed from rank 0 to the others over a plain TCP socket on MASTER_PORT+1.. (the launcher's
own store owns MASTER_PORT); no torch import is needed.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import socket
import struct
import time

from . import _lib
from .device import get_device

_MAGIC = b"MSKRDZV1"
_PORT_SPAN = 16


class ParallelEnv:
    """paddle.distributed.ParallelEnv look-alike (nranks, local_rank, rank)."""

    @property
    def nranks(self):
        return int(os.environ.get("WORLD_SIZE", "1"))

    world_size = nranks

    @property
    def rank(self):
        return int(os.environ.get("RANK", "0"))

    @property
    def local_rank(self):
        return int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


def _job_token(world, channel=0):
    s = "{}:{}:{}:{}".format(os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"), world, channel)
    return hashlib.sha256(s.encode()).digest()[:16]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed during rendezvous")
        buf += chunk
    return buf


def exchange_bytes(payload: bytes | None, rank: int, world: int, timeout=300.0, channel=None) -> bytes:
    """Rank 0 broadcasts ``payload`` to all other ranks (used for the RCCL unique id).
    Protocol: client sends MAGIC+token+rank; server answers MAGIC+len+payload.
    channel (default: $MSEGK_RDZV_CHANNEL or 0): independent rendezvous of the same job use disjoint port ranges and tokens
    (MASTER_PORT + 1 + 16 * channel ...) -- launch.py gives every relaunched attempt and the supervisors their own."""
    if world == 1:
        return payload
    if channel is None:
        channel = int(os.environ.get("MSEGK_RDZV_CHANNEL", "0"))
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(os.environ.get("MASTER_PORT", "29500")) + 1 + _PORT_SPAN * int(channel)
    token = _job_token(world, channel)
    if rank == 0:
        srv = None
        for port in range(base, base + _PORT_SPAN):
            try:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                s.bind((addr if addr not in ("localhost", ) else "127.0.0.1", port))
                s.listen(world)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError("rendezvous: no free port in [{}, {})".format(base, base + _PORT_SPAN))
        srv.settimeout(timeout)
        served = set()
        while len(served) < world - 1:
            conn, _ = srv.accept()
            try:
                conn.settimeout(10.0)
                hello = _recv_exact(conn, len(_MAGIC) + 16 + 4)
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):len(_MAGIC) + 16] != token:
                    conn.close()
                    continue
                peer = struct.unpack("<i", hello[-4:])[0]
                conn.sendall(_MAGIC + struct.pack("<i", len(payload)) + payload)
                served.add(peer)
            except (OSError, ConnectionError):
                pass
            finally:
                conn.close()
        srv.close()
        return payload
    deadline = time.time() + timeout
    while time.time() < deadline:
        for port in range(base, base + _PORT_SPAN):
            try:
                with socket.create_connection((addr, port), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(_MAGIC + token + struct.pack("<i", rank))
                    head = _recv_exact(conn, len(_MAGIC) + 4)
                    if head[:len(_MAGIC)] != _MAGIC:
                        continue
                    n = struct.unpack("<i", head[-4:])[0]
                    return _recv_exact(conn, n)
            except (OSError, ConnectionError):
                continue
        time.sleep(0.2)
    raise TimeoutError("rendezvous with rank 0 timed out")


def spawn_ranks(n, argv=None, env=None, timeout=None):
    """Self-launch: run `argv` (default: this process's own command line) as n ranks of ONE node, one process per GPU --
    what `python -m torch.distributed.run --nproc-per-node n` would do for this package's launch contract, without torch:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT=<a free port> in each child's environment (the
    RCCL id then travels over `exchange_bytes`).  The ranks run under launch.run_supervised: a rank that fails OR hangs
    (no heartbeat within the phase limits of launch.py; workers that never call launch.heartbeat are only bounded by
    `timeout`) takes the others down; SIGINT / SIGTERM and the death of this process kill the ranks.  Rank 0's stdout is
    this process's stdout.  Returns 0, or 1 / 124 (failed / timed out).
    Replaces `python -m paddle.distributed.launch train.py ...` (reference README / run-vnet.sh) for train.py when no
    external launcher set WORLD_SIZE (bench.py drives launch.run_supervised itself, with fall-back plans)."""
    import sys
    from . import launch
    argv = list(argv) if argv is not None else [sys.executable] + sys.argv
    base = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        base.pop(k, None)
    base.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()), MSEGK_SELF_LAUNCHED="1")
    # a generic worker sends no heartbeats: only `timeout` (or MSEGK_WATCHDOG_S) bounds it
    wd = None if os.environ.get("MSEGK_WATCHDOG_S") else (timeout if timeout is not None else 10 ** 9)
    ok, text, attempts = launch.run_supervised(argv, [{"label": "as launched", "extra": []}], n, range(n), env=base,
                                               total_timeout=timeout, watchdog_s=wd)
    if text:
        sys.stdout.write(text)
        sys.stdout.flush()
    if ok:
        return 0
    return 124 if attempts and attempts[-1]["outcome"] == "hang" else 1


_initialised = False
DRY_RUN = False     # bench.py's compute-only replay: True = no collective is issued (BatchNorm statistics stay rank-local,
                    # the gradient exchange is skipped) -- what the step costs WITHOUT communication, for dp.exposed_comm_ms_per_step


def set_dry_run(on):
    """Compute-only replay for bench.py's dp.exposed_comm_ms_per_step: while on, neither the SyncBatchNorm exchanges nor the
    gradient all-reduce / buckets are issued (every rank runs the step on its own data with rank-local statistics)."""
    global DRY_RUN
    from . import nn
    DRY_RUN = bool(on)
    nn.BatchNorm3D.dry_run = bool(on)


def resolve_dp_mode(dp_mode, world):
    """'auto' (bench.py / train.py --dp-mode auto, the default): gradient buckets overlapped with backward on a second
    communicator + stream (mode 2) whenever there is more than one rank -- the exposed-communication budget for >= 6.5x at
    8 GPUs is 19.3 ms x (8 / 6.5 - 1) = 4.45 ms per step and mode 0 leaves the whole 182 MB all-reduce exposed; msk_dp_init
    falls back to mode 0 (and says so on stderr) when ncclCommSplit cannot give it the second communicator.  None = leave
    the library's current setting (0 unless MSEGK_DP_MODE / msk_set_option changed it)."""
    if dp_mode is None:
        return None
    if isinstance(dp_mode, str):
        if dp_mode == "auto":
            return 2 if world > 1 else 0
        dp_mode = int(dp_mode)
    if dp_mode not in (0, 1, 2, 3):
        raise ValueError("dp_mode must be 'auto', 0, 1, 2 or 3, got %r" % (dp_mode,))
    return dp_mode


def rccl_version():
    """'2.27.3' from ncclGetVersion's 22703 (RCCL's own numbering), or None"""
    v = C.c_int(0)
    if _lib.load().msk_dp_rccl_version(C.byref(v)) != 0 or v.value <= 0:
        return None
    n = v.value
    return "%d.%d.%d" % (n // 10000, (n // 100) % 100, n % 100) if n >= 10000 else "%d.%d.%d" % (n // 1000, (n // 100) % 10, n % 100)


def binding_info(env=None, dev=None):
    """What a multi-rank run logs about where it runs (bench.py dp.binding): the device this rank is bound to, how many are
    visible, the IPC mode RCCL will use, the RCCL version."""
    env = env or ParallelEnv()
    dev = dev or get_device()
    n = C.c_int(0)
    _lib.load().msk_device_count(C.byref(n))
    return {"rank": env.rank, "local_rank": env.local_rank, "device_index": dev.index, "device": dev.name(),
            "pci": dev.pci_bus_id(), "visible_devices": n.value,
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "rccl_version": rccl_version()}


def init_parallel_env(dp_mode=None):
    """Create the RCCL communicator for this process's device (idempotent).  dp_mode (msk_dp.hip; train.py / bench.py
    --dp-mode): 0 = every collective on the compute stream (default), 2 = gradient buckets on a second communicator and
    stream, overlapped with backward (DataParallel then defaults to overlap=True); the environment variable MSEGK_DP_MODE
    still overrides it inside msk_dp_init."""
    global _initialised
    env = ParallelEnv()
    dev = get_device()
    dp_mode = resolve_dp_mode(dp_mode, env.nranks)
    if _initialised or env.nranks == 1:
        # the arrangement is fixed when the communicators are created (mode 2's second one): a later request cannot take
        # effect -- say so instead of ignoring it (advisor, round 4)
        if dp_mode is not None and _initialised and dp_mode != dev.get_option("dp_mode"):
            import warnings
            warnings.warn("init_parallel_env(dp_mode=%d) after the communicators were created: dp_mode stays %d"
                          % (dp_mode, dev.get_option("dp_mode")))
        return env
    if dp_mode is not None:
        dev.set_option("dp_mode", int(dp_mode))
    lib = _lib.load()
    # the rank -> GPU binding, checked (device.local_device_index) and logged BEFORE the communicator exists
    info = binding_info(env, dev)
    import sys
    sys.stderr.write("[msegk] rank %d/%d: LOCAL_RANK %d -> HIP device %d (%s, PCI %s), %d device(s) visible, "
                     "HSA_ENABLE_IPC_MODE_LEGACY=%s, RCCL %s\n"
                     % (env.rank, env.nranks, env.local_rank, dev.index, info["device"], info["pci"], info["visible_devices"],
                        info["HSA_ENABLE_IPC_MODE_LEGACY"], info["rccl_version"]))
    uid = None
    if env.rank == 0:
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        if lib.msk_dp_unique_id(buf) != 0:
            raise _lib.MskError("msk_dp_unique_id failed: " + _lib.last_error(None))
        uid = buf.raw
    uid = exchange_bytes(uid, env.rank, env.nranks)
    try:
        dev.call("msk_dp_init", uid, env.rank, env.nranks)
    except _lib.MskError as e:
        from .device import _MASK_VARS
        masks = {v: os.environ[v] for v in _MASK_VARS if os.environ.get(v)}
        raise _lib.MskError("%s -- rank %d of %d: LOCAL_RANK %d -> HIP device %d (PCI %s), %d device(s) visible, masks %s.  RCCL "
                            "reports 'invalid usage' when two ranks are bound to the same GPU (fewer visible devices than ranks "
                            "per node, or one *_VISIBLE_DEVICES mask shared by all ranks)"
                            % (e, env.rank, env.nranks, env.local_rank, dev.index, info["pci"], info["visible_devices"],
                               masks or "none")) from None
    dev.rank, dev.world = env.rank, env.nranks
    _initialised = True
    return env


def barrier():
    dev = get_device()
    if dev.world > 1:
        dev.call("msk_dp_barrier")
    else:
        dev.sync()


class DataParallel:
    """paddle.DataParallel / fleet.distributed_model replacement.

    * parameters and BN buffers are broadcast from rank 0 at wrap time (App. B.9);
    * the flat gradient arena (182.4 MB for VNet) is summed in BUCKETS while backward is still running: the model
      reports every block whose backward has been enqueued (`_grad_ready_hooks`); as soon as the finished blocks form
      a contiguous tail of the arena of at least `bucket_bytes`, that slice goes to `msk_dp_allreduce_async` (second
      communicator, communication stream).  Whatever is left is sent after the last block, then the compute stream
      waits for all buckets (`msk_dp_wait`) before the optimizer, which applies 1/nranks;
    * BatchNorm statistics are exchanged inside the layers (SyncBatchNorm semantics).

    Default (`overlap=None`): ONE all-reduce of the arena after backward, every collective (the SyncBatchNorm exchanges
    too) on the compute stream's single communicator -- `msk_dp.hip` dp_mode 0.  Round 3 built and MEASURED the alternatives
    on a 1-rank RCCL communicator with the 48 statistics collectives forced on (plain step 20.7 ms): this default 21.3 ms;
    buckets overlapped on a second communicator + stream (`MSEGK_DP_MODE=2 MSEGK_DP_OVERLAP=1`, or `overlap=True`) 21.5 ms;
    the single-communicator / single-communication-stream arrangement the round-2 review proposed (`MSEGK_DP_MODE=1`) is
    correct but costs 42.6 ms per step, its variant with only the buckets on the second stream (`MSEGK_DP_MODE=3`) 35.4 ms
    (details in `msk_dp.hip`).  Exposed cost of the default at 8 GPUs: the 182 MB all-reduce, ~1.3-2 ms."""

    def __init__(self, model, overlap=None, bucket_bytes=16 << 20, force=False):
        self._layers = model
        dev = model.dev
        self.dev = dev
        self.bucket_elems = max(1, int(bucket_bytes) // 4)
        self.buckets_last_step = []      # [(offset, count)] of the previous backward, for tests / logs
        if dev.world > 1 or force:       # force: exercise the hooks on a 1-rank communicator (tests)
            dev.call("msk_dp_broadcast", C.c_void_p(model.arena.value_ptr), C.c_size_t(model.arena.count), 0)
            if model.buffer_arena.count:
                dev.call("msk_dp_broadcast", C.c_void_p(model.buffer_arena.value_ptr),
                         C.c_size_t(model.buffer_arena.count), 0)
            fa = getattr(model, "frozen_arena", None)
            if fa is not None and fa.count:
                dev.call("msk_dp_broadcast", C.c_void_p(fa.value_ptr), C.c_size_t(fa.count), 0)
            model.arena.grad_scale = 1.0 / dev.world
            mode = dev.get_option("dp_mode")          # the EFFECTIVE arrangement (option, MSEGK_DP_MODE, fall-backs of msk_dp_init)
            if overlap is None:      # an explicit argument wins over the environment (advisor finding, round 2)
                env_o = os.environ.get("MSEGK_DP_OVERLAP")
                overlap = (env_o != "0") if env_o is not None else mode != 0
            if overlap and mode == 0:
                # in mode 0 msk_dp_allreduce_async IS msk_dp_allreduce_sum on the compute stream: every bucket would join the
                # weight-gradient stream and serialise backward -- slower than one all-reduce after backward (advisor, round 3)
                import warnings
                warnings.warn("DataParallel(overlap=True) needs dp_mode 1-3 (train.py / bench.py --dp-mode 2, or MSEGK_DP_MODE=2 "
                              "before init_parallel_env); dp_mode is 0: falling back to ONE all-reduce after backward.")
                overlap = False
            self.overlap = bool(overlap and hasattr(model, "_grad_ready_hooks"))   # what really runs (a model without block hooks: one all-reduce)
            if self.overlap:
                params = model.arena.params
                self._index = {id(p): i for i, p in enumerate(params)}
                self._start = [p.offset for p in params] + [model.arena.count]
                self._done = [False] * len(params)
                self._tail = len(params)     # params[_tail:] are already on the wire
                self._members = {}
                self._sent = []
                model._grad_ready_hooks.append(self._block_ready)
                model._post_backward_hooks.append(self._finish)
            else:
                model._post_backward_hooks.append(self._allreduce)

    # -- single all-reduce (overlap=False) ------------------------------------------------
    def _allreduce(self, model):
        a = model.arena
        if DRY_RUN:
            self.buckets_last_step = []
            return
        self.dev.call("msk_dp_allreduce_sum", C.c_void_p(a.grad_ptr), C.c_size_t(a.count))
        self.buckets_last_step = [(0, a.count)]

    # -- bucketed, overlapped --------------------------------------------------------------
    def _send(self, lo_idx, hi_idx):
        a = self._layers.arena
        lo, hi = self._start[lo_idx], self._start[hi_idx]
        if hi > lo and not DRY_RUN:
            self.dev.call("msk_dp_allreduce_async", C.c_void_p(a.grad_ptr + 4 * lo), C.c_size_t(hi - lo))
            self._sent.append((lo, hi - lo))
        self._tail = lo_idx

    def _block_ready(self, model, block):
        idx = self._members.get(id(block))
        if idx is None:
            idx = [self._index[id(p)] for p in block.parameters() if id(p) in self._index]
            self._members[id(block)] = idx
        for i in idx:
            self._done[i] = True
        lo = self._tail
        while lo > 0 and self._done[lo - 1]:
            lo -= 1
        if lo < self._tail and (self._start[self._tail] - self._start[lo] >= self.bucket_elems or lo == 0):
            self._send(lo, self._tail)

    def _finish(self, model):
        if self._tail > 0:               # everything is enqueued by now, finished-or-not bookkeeping aside
            self._send(0, self._tail)
        self.dev.call("msk_dp_wait")
        self.buckets_last_step, self._sent = self._sent, []
        self._done = [False] * len(self._done)
        self._tail = len(self._done)

    def __call__(self, *args, **kw):
        return self._layers(*args, **kw)

    def __getattr__(self, name):
        return getattr(self._layers, name)


def shard_indices(n_items: int, batch_size: int, rank: int, world: int, shuffle: bool, epoch: int, seed=0,
                  drop_last=False):
    """paddle.io.DistributedBatchSampler semantics (core/train.py:87-88): one shared
    permutation per epoch, padded so every rank sees the same number of samples, rank r
    takes the r-th contiguous chunk of ceil(n/world) indices; yields lists of batch indices."""
    import numpy as np
    idx = np.arange(n_items)
    if shuffle:
        np.random.RandomState(seed + epoch).shuffle(idx)
    idx = idx.tolist()
    per = (n_items + world - 1) // world
    total = per * world
    idx += idx[:total - len(idx)]
    # full blocks of world*bs indices are dealt batch-wise (rank r takes the r-th batch of each
    # block); the ragged tail (< world*bs indices, a multiple of world) is cut into `world` equal
    # contiguous pieces -- every rank ends up with exactly `per` samples
    block = batch_size * world
    tail = total % block
    mine = []
    for start in range(rank * batch_size, total - tail, block):
        mine.extend(idx[start:start + batch_size])
    tail_idx = idx[total - tail:]
    piece = tail // world
    mine.extend(tail_idx[rank * piece:(rank + 1) * piece])
    batches = [mine[i:i + batch_size] for i in range(0, len(mine), batch_size)]
    if drop_last and batches and len(batches[-1]) < batch_size:
        batches.pop()
    return batches
