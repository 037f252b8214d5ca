"""Supervised launch of the ranks of ONE node: a hang watchdog, a bounded list of fall-back plans, and a result on every outcome.

Replaces `python -m paddle.distributed.launch train.py ...` (reference README / run-vnet.sh:12-16) and the failure handling
paddle's launcher does for `core/train.py:81-95` (fleet.init / distributed_model): a rank that dies takes the job down, and here
additionally a rank that HANGS (a collective that never completes, a peer that never reaches the rendezvous) is detected
from the outside, because the hung process cannot report it itself.

Two shapes, one code path (`run_supervised`):
  * self-launched (`python bench.py --gpus N`, WORLD_SIZE unset): this process supervises all N workers;
  * under an external launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): every launched rank
    is the supervisor of ONE worker on its own GPU; the supervisors agree on the outcome of an attempt through small status
    files in a shared directory whose name rank 0 hands out over the TCP rendezvous (`parallel.exchange_bytes`).

Worker side: `heartbeat(phase)` appends one line to $MSEGK_HEARTBEAT_FILE; the supervisor reads the file's last line and
age.  The limit that applies is the one of the phase reached LAST (how long the NEXT one may take to arrive).
No torch, no GPU call in this module.
"""
from __future__ import annotations

import collections
import json
import os
import signal
import subprocess
import sys
import tempfile
import threading
import time

# seconds the NEXT heartbeat may take after the named phase was reached ("start" = process spawned, nothing reported yet: the
# interpreter + library load of a fresh box; "import" -> RCCL communicator creation; everything later is one step or one
# short untimed pass of the job).  MSEGK_WATCHDOG_S overrides all of them (tests), MSEGK_WATCHDOG_STEP_S the default.
PHASE_LIMITS = {"start": 300.0, "import": 240.0, "dp_init": 120.0}
DEFAULT_LIMIT = 120.0
PEER_GRACE_S = 60.0          # how long a supervisor waits for the status files of the OTHER supervisors of an attempt
SUPERVISOR_CHANNEL = 7       # parallel.exchange_bytes channel of the supervisors' own rendezvous
ATTEMPT_CHANNEL0 = 3         # workers of attempt k rendezvous on channel ATTEMPT_CHANNEL0 + k


def heartbeat(phase):
    """Worker side: report that `phase` has been reached (a no-op when the process is not supervised)."""
    path = os.environ.get("MSEGK_HEARTBEAT_FILE")
    if not path:
        return
    try:
        with open(path, "a") as f:
            f.write("%s\t%.3f\n" % (phase, time.time()))
    except OSError:
        pass


def _limit_after(phase, override=None):
    if override is not None:
        return float(override)
    env_all = os.environ.get("MSEGK_WATCHDOG_S")
    if env_all:
        return float(env_all)
    base = phase.split(" ")[0] if phase else "start"
    if base in PHASE_LIMITS:
        return PHASE_LIMITS[base]
    return float(os.environ.get("MSEGK_WATCHDOG_STEP_S", DEFAULT_LIMIT))


def _write_status(d, attempt, rank, rec):
    tmp = os.path.join(d, ".a%d.r%d.tmp" % (attempt, rank))
    with open(tmp, "w") as f:
        json.dump(rec, f)
    os.replace(tmp, os.path.join(d, "a%d.r%d.status" % (attempt, rank)))


def _read_statuses(d, attempt, world):
    out = {}
    for r in range(world):
        p = os.path.join(d, "a%d.r%d.status" % (attempt, r))
        try:
            with open(p) as f:
                out[r] = json.load(f)
        except (OSError, ValueError):
            pass
    return out


def _set_pdeathsig():
    # the worker must not outlive its supervisor (a killed launcher would otherwise leave ranks holding GPUs)
    try:
        import ctypes
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG
    except Exception:
        pass


class _Child:
    def __init__(self, rank, argv, env, hb_path, forward_stdout):
        self.rank = rank
        self.hb_path = hb_path
        self.t0 = time.time()
        self.out_lines = []
        self.err_tail = collections.deque(maxlen=40)
        self.status = None
        open(hb_path, "w").close()
        self.proc = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1,
                                     start_new_session=True, preexec_fn=_set_pdeathsig)
        self._threads = [threading.Thread(target=self._pump_out, args=(forward_stdout,), daemon=True),
                         threading.Thread(target=self._pump_err, daemon=True)]
        for t in self._threads:
            t.start()

    def _pump_out(self, forward):
        for line in self.proc.stdout:
            self.out_lines.append(line)
            if forward:                      # not the result rank: its stdout is diagnostics
                sys.stderr.write(line)
        self.proc.stdout.close()

    def _pump_err(self):
        for line in self.proc.stderr:
            self.err_tail.append(line.rstrip("\n"))
            sys.stderr.write(line)
        self.proc.stderr.close()

    def last_phase(self):
        """(phase reached last, seconds since then)"""
        try:
            with open(self.hb_path) as f:
                lines = f.read().splitlines()
        except OSError:
            lines = []
        if not lines:
            return "start", time.time() - self.t0
        ph, ts = lines[-1].rsplit("\t", 1)
        return ph, time.time() - float(ts)

    def kill(self):
        if self.proc.poll() is not None:
            return
        for sig, wait in ((signal.SIGTERM, 3.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(self.proc.pid, sig)          # the worker's whole session (start_new_session)
            except (ProcessLookupError, PermissionError):
                try:
                    self.proc.send_signal(sig)
                except ProcessLookupError:
                    pass
            try:
                self.proc.wait(timeout=wait)
                return
            except subprocess.TimeoutExpired:
                continue

    def join_output(self):
        for t in self._threads:
            t.join(timeout=2.0)


def shared_dir(rank, world, owns_all):
    """The directory the supervisors of a job write their status files into (rank 0 creates it and hands the name out)."""
    if owns_all:
        return tempfile.mkdtemp(prefix="msegk_sup_")
    from . import parallel
    d = tempfile.mkdtemp(prefix="msegk_sup_") if rank == 0 else None
    got = parallel.exchange_bytes(d.encode() if d else None, rank, world, timeout=_limit_after("start"),
                                  channel=SUPERVISOR_CHANNEL)
    return got.decode()


def run_supervised(argv, plans, world, ranks, env=None, result_rank=0, total_timeout=None, rdv_dir=None, watchdog_s=None):
    """Run `argv + plan["extra"]` as the worker of every rank in `ranks` (this supervisor's share of range(world)); on a
    failed or hung attempt kill the workers and go to the next plan.

    plans: [{"label": str, "extra": [argv...], "env": {..}}] -- tried in order, at most len(plans) attempts.
    Returns (ok, result_text, attempts): result_text = the stdout of `result_rank`'s worker in the successful attempt (None
    when this supervisor does not own that rank or nothing succeeded); attempts = [{"plan", "outcome", "seconds", ...}].
    watchdog_s: one heartbeat limit for every phase (workers that send no heartbeats: pass the job's own time limit)."""
    ranks = list(ranks)
    owns_all = sorted(ranks) == list(range(world))
    base_env = dict(os.environ if env is None else env)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL across processes)
    d = rdv_dir or shared_dir(ranks[0], world, owns_all)
    t_job = time.time()
    attempts = []
    children = []

    def _on_signal(signum, frame):
        for c in children:
            c.kill()
        raise SystemExit(128 + signum)

    old = {}
    if threading.current_thread() is threading.main_thread():
        for s in (signal.SIGINT, signal.SIGTERM):
            old[s] = signal.signal(s, _on_signal)
    try:
        for k, plan in enumerate(plans):
            t_att = time.time()
            children = []
            for r in ranks:
                e = dict(base_env)
                e.update(plan.get("env", {}))
                e.update(WORLD_SIZE=str(world), RANK=str(r), MSEGK_SUPERVISED="1", MSEGK_ATTEMPT=str(k),
                         MSEGK_RDZV_CHANNEL=str(ATTEMPT_CHANNEL0 + k),
                         MSEGK_HEARTBEAT_FILE=os.path.join(d, "a%d.r%d.hb" % (k, r)))
                if owns_all:
                    e["LOCAL_RANK"] = str(r)
                children.append(_Child(r, list(argv) + list(plan.get("extra", [])), e, e["MSEGK_HEARTBEAT_FILE"],
                                       forward_stdout=(r != result_rank)))
            abort_seen = None
            t_mine_done = None
            while True:
                for c in children:
                    if c.status is not None:
                        continue
                    code = c.proc.poll()
                    ph, age = c.last_phase()
                    if code is not None:
                        c.join_output()
                        c.status = {"state": "ok" if code == 0 else "fail", "rc": code, "phase": ph,
                                    "seconds": round(time.time() - c.t0, 1)}
                        if code != 0:
                            c.status["error"] = "\n".join(list(c.err_tail)[-8:])[-1500:]
                    elif age > _limit_after(ph, watchdog_s):
                        c.status = {"state": "hang", "phase": ph, "stale_s": round(age, 1), "limit_s": _limit_after(ph, watchdog_s),
                                    "seconds": round(time.time() - c.t0, 1),
                                    "error": "no heartbeat for %.0f s after phase %r" % (age, ph)}
                        c.kill()
                    elif total_timeout is not None and time.time() - t_job > total_timeout:
                        c.status = {"state": "hang", "phase": ph, "seconds": round(time.time() - c.t0, 1),
                                    "error": "job exceeded its total limit of %.0f s" % total_timeout}
                        c.kill()
                    if c.status is not None:
                        _write_status(d, k, c.rank, c.status)
                st = _read_statuses(d, k, world)
                bad = [r for r, s in sorted(st.items()) if s["state"] in ("fail", "hang")]
                if bad and abort_seen is None:
                    abort_seen = bad[0]
                    for c in children:             # a failed / hung rank takes the attempt down: its peers wait in a collective
                        if c.status is None:
                            ph, _ = c.last_phase()
                            c.kill()
                            c.status = {"state": "aborted", "phase": ph, "because_rank": abort_seen,
                                        "seconds": round(time.time() - c.t0, 1)}
                            _write_status(d, k, c.rank, c.status)
                if all(c.status is not None for c in children):
                    if t_mine_done is None:
                        t_mine_done = time.time()
                    if len(st) == world or time.time() - t_mine_done > PEER_GRACE_S:
                        break
                time.sleep(0.05)
            st = _read_statuses(d, k, world)
            lost = [r for r in range(world) if r not in st]
            ok = not lost and all(s["state"] == "ok" for s in st.values())
            rec = {"plan": plan.get("label", "plan %d" % k), "extra_args": list(plan.get("extra", [])),
                   "outcome": "ok" if ok else "failed", "seconds": round(time.time() - t_att, 1)}
            if not ok:
                culprit = next((r for r in sorted(st) if st[r]["state"] in ("hang", "fail")), None)
                if culprit is not None:
                    s = st[culprit]
                    rec.update(outcome=s["state"], rank=culprit, phase=s.get("phase"), error=s.get("error"), rc=s.get("rc"))
                elif lost:
                    rec.update(outcome="lost", error="no status from the supervisor(s) of rank(s) %s" % lost)
                rec["phases"] = {str(r): st[r].get("phase") for r in sorted(st)}
            attempts.append(rec)
            if ok:
                mine = next((c for c in children if c.rank == result_rank), None)
                return True, ("".join(mine.out_lines) if mine else None), attempts
            sys.stderr.write("[msegk launch] attempt %d (%s) %s at rank %s, phase %r%s\n"
                             % (k, rec["plan"], rec["outcome"], rec.get("rank"), rec.get("phase"),
                                ": next plan" if k + 1 < len(plans) else ": no plan left"))
            if total_timeout is not None and time.time() - t_job > total_timeout:
                break
            if k + 1 < len(plans):      # the killed workers' GPU contexts and RCCL resources go away asynchronously
                time.sleep(float(os.environ.get("MSEGK_RELAUNCH_DELAY_S", "1.0")))
        return False, None, attempts
    finally:
        for c in children:
            c.kill()
        for s, h in old.items():
            signal.signal(s, h)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_context(n):
    """(world, ranks, env) for a supervisor: under an external launcher (WORLD_SIZE set) this process owns its own RANK; with
    no launcher it owns all n ranks of 127.0.0.1 and picks the rendezvous port."""
    env = dict(os.environ)
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        return world, [int(env.get("RANK", "0"))], env
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), MSEGK_SELF_LAUNCHED="1")
    return n, list(range(n)), env
