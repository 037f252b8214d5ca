"""`python bench.py --gpus N` must come back with ONE diagnosable JSON line whatever the ranks do (round-5 verdict, item 1):
a rank that never reaches the rendezvous, a rank that hangs in its first collective, a rank that exits non-zero -- over the
no-compute stand-in library (tests/fake_msegk.c, fault injection by environment), self-launched and under the driver's
launcher (`python -m torch.distributed.run`).  The supervisor is medicalseg_amd/launch.py; the fall-back plans are bench.py's
(`--dp-mode 0`, then `--dp-mode 0 --no-sync-bn`).  Replaces the failure handling of paddle.distributed.launch around
reference core/train.py:81-95."""
import json
import os
import subprocess
import sys
import time

import pytest

from test_bench_launch_dryrun import _fake_lib, _free_port

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = [os.path.join(HERE, "run_bench_fake.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "16", "--no-cpu-baseline"]


def _env(tmp_path, **kw):
    env = dict(os.environ, MSK_FAKE_LIB=_fake_lib(tmp_path), OMP_NUM_THREADS="1", MSEGK_WATCHDOG_S="10")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FAKE_FAULT"):
        env.pop(k, None)
    env.update({k: str(v) for k, v in kw.items()})
    return env


def _run(cmd, env, limit):
    t0 = time.time()
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=limit)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-3000:])      # exactly ONE JSON line, on every outcome
    return out.returncode, json.loads(lines[0]), time.time() - t0


def test_rank_hangs_in_first_collective_relaunches_in_dp_mode_0(tmp_path):
    rc, rec, dt = _run([sys.executable] + BENCH, _env(tmp_path, FAKE_FAULT="hang_collective", FAKE_FAULT_DP_MODE=2), 120)
    att = rec["dp"]["attempts"]
    assert rc == 0 and rec["value"] > 0 and rec["n_gpus"] == 2
    assert [a["outcome"] for a in att] == ["hang", "ok"] and att[0]["rank"] == 1 and att[1]["extra_args"] == ["--dp-mode", "0"]
    assert att[0]["phase"] == "model" and "no heartbeat" in att[0]["error"]       # hung before its first warm-up step completed
    assert rec["dp"]["dp_mode"] == 0 and rec["config"]["sync_bn"] is True          # the number is the fall-back's, and says so
    assert dt < 60


def test_rank_hangs_in_every_arrangement_with_sync_bn_ends_with_local_statistics(tmp_path):
    # the statistics exchange itself hangs (whatever the arrangement): the third plan drops it -- a deviation the line reports
    env = _env(tmp_path, FAKE_FAULT="hang_collective", FAKE_FAULT_LAST_ATTEMPT=1)
    rc, rec, dt = _run([sys.executable] + BENCH, env, 180)
    att = rec["dp"]["attempts"]
    assert rc == 0 and [a["outcome"] for a in att] == ["hang", "hang", "ok"]
    assert att[2]["extra_args"] == ["--dp-mode", "0", "--no-sync-bn"] and rec["config"]["sync_bn"] is False


def test_rank_exits_nonzero_relaunches_and_reports_the_failure(tmp_path):
    rc, rec, dt = _run([sys.executable] + BENCH, _env(tmp_path, FAKE_FAULT="exit_collective", FAKE_FAULT_LAST_ATTEMPT=0), 120)
    att = rec["dp"]["attempts"]
    assert rc == 0 and rec["value"] > 0
    assert [a["outcome"] for a in att] == ["fail", "ok"] and att[0]["rc"] == 3 and att[0]["rank"] == 1


def test_rank_never_reaches_the_rendezvous_yields_a_failure_line(tmp_path):
    rc, rec, dt = _run([sys.executable] + BENCH, _env(tmp_path, FAKE_FAULT="hang_ctx", MSEGK_WATCHDOG_S=6), 180)
    assert rc != 0 and rec["value"] is None and rec["n_gpus"] == 2 and rec["unit"] == "voxels/s"
    att = rec["dp"]["attempts"]
    assert len(att) == 3 and all(a["outcome"] == "hang" and a["rank"] == 1 and a["phase"] == "start" for a in att)
    assert "no attempt produced a result" in rec["error"] and "rank 1" in rec["error"]
    assert att[0]["phases"] == {"0": "import", "1": "start"}      # rank 0 was waiting in the rendezvous
    assert dt < 90


def test_every_rank_fails_yields_a_failure_line_with_the_error_text(tmp_path):
    # an unknown --opt key raises in every rank (ValueError from `k, v = kv.split("=")`): three failed attempts, one line
    rc, rec, dt = _run([sys.executable] + BENCH + ["--opt", "broken"], _env(tmp_path), 180)
    assert rc != 0 and rec["value"] is None
    assert [a["outcome"] for a in rec["dp"]["attempts"]] == ["fail"] * 3 and "ValueError" in rec["error"]


def test_under_the_drivers_launcher_a_hang_is_relaunched_by_the_per_rank_supervisors(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: each launched rank supervises ONE worker; the two
    supervisors agree through the shared status directory, so rank 0's worker (which did not hang) is relaunched too."""
    env = _env(tmp_path, FAKE_FAULT="hang_collective", FAKE_FAULT_DP_MODE=2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + BENCH
    rc, rec, dt = _run(cmd, env, 300)
    att = rec["dp"]["attempts"]
    assert rc == 0 and rec["value"] > 0 and rec["dp"]["launcher"].startswith("external")
    assert [a["outcome"] for a in att] == ["hang", "ok"] and att[0]["rank"] == 1
    assert rec["dp"]["dp_mode"] == 0


def test_launcher_with_the_wrong_world_size_yields_a_failure_line(tmp_path):
    env = _env(tmp_path, WORLD_SIZE=4, RANK=0, LOCAL_RANK=0, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    rc, rec, dt = _run([sys.executable] + BENCH, env, 60)
    assert rc != 0 and rec["value"] is None and "WORLD_SIZE=4" in rec["error"]


def test_local_rank_to_device_binding_is_checked():
    from medicalseg_amd._lib import MskError
    from medicalseg_amd.device import local_device_index
    assert local_device_index(3, 8, {}) == 3
    assert local_device_index(0, 1, {}) == 0
    assert local_device_index(5, 1, {"HIP_VISIBLE_DEVICES": "5"}) == 0        # the launcher masked one GPU per rank
    with pytest.raises(MskError, match="LOCAL_RANK=2 but only 2"):
        local_device_index(2, 2, {})
    with pytest.raises(MskError):
        local_device_index(1, 1, {})                                           # two ranks would share GPU 0 silently
    assert local_device_index(1, 0, {}) == 1                                   # no device at all: Device() raises "no HIP device"


def test_heartbeat_limits_follow_the_phase_reached_last(monkeypatch):
    from medicalseg_amd import launch
    monkeypatch.delenv("MSEGK_WATCHDOG_S", raising=False)
    monkeypatch.delenv("MSEGK_WATCHDOG_STEP_S", raising=False)
    assert launch._limit_after("start") == 300 and launch._limit_after("import") == 240 and launch._limit_after("dp_init") == 120
    assert launch._limit_after("warmup 3") == launch.DEFAULT_LIMIT
    monkeypatch.setenv("MSEGK_WATCHDOG_STEP_S", "45")
    assert launch._limit_after("model") == 45 and launch._limit_after("start") == 300
    monkeypatch.setenv("MSEGK_WATCHDOG_S", "7")
    assert launch._limit_after("start") == 7 and launch._limit_after("timed") == 7
    assert launch._limit_after("start", 99) == 99
