"""bench.py under the driver's launcher (`python -m torch.distributed.run --nproc-per-node 2 ...`)
with the no-compute stand-in library: proves the N > 1 launch contract end to end on CPU --
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* parsing, the unique-id rendezvous next to the launcher's own
store, barrier/max-over-ranks plumbing, exactly ONE JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_lib(tmp_path):
    so = str(tmp_path / "libfake_msegk.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-o", so, os.path.join(HERE, "fake_msegk.c")])
    return so


def test_bench_two_ranks_prints_one_json_line(tmp_path):
    env = dict(os.environ, MSK_FAKE_LIB=_fake_lib(tmp_path), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "run_bench_fake.py"), "--gpus", "2",
           "--steps", "2", "--warmup", "1", "--size", "16", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["global_batch"] == 4 and rec["config"]["parallelism"] == "dp2"
    assert rec["unit"] == "voxels/s" and rec["higher_is_better"] is True and "roofline" in rec
    assert "cpu_baseline" not in rec                         # rank 0 at N=1 only
    # round 4: the first multi-GPU run must be diagnosable from the line alone
    dp = rec["dp"]
    assert dp["dp_mode"] == 0 and dp["overlap_buckets"] is False and dp["buckets_last_step"] == 1
    assert len(dp["allreduce_busbw_GBps_per_rank"]) == 2 and len(dp["syncbn_collective_ms_per_step_per_rank"]) == 2
    assert set(dp["per_rank_collective_ms_per_step"]) == {"rccl_allreduce", "rccl_allreduce_stats", "rccl_allgather", "rccl_allreduce_bucket"}
    assert rec["roofline"]["strict_fp32"] is None            # the exact-operand pass is a 1-GPU extra
    # --dp-mode 2: gradient buckets on the communication stream, overlapped with backward
    cmd2 = cmd[:cmd.index(os.path.join(HERE, "run_bench_fake.py"))]
    cmd2[cmd2.index("--master-port") + 1] = str(_free_port())
    cmd2 += [os.path.join(HERE, "run_bench_fake.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "16", "--no-cpu-baseline",
             "--dp-mode", "2"]
    out2 = subprocess.run(cmd2, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    rec2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith("{")][0])
    assert rec2["dp"]["dp_mode"] == 2 and rec2["dp"]["overlap_buckets"] is True and rec2["dp"]["buckets_last_step"] >= 2


def test_bench_single_process_defaults(tmp_path):
    env = dict(os.environ, MSK_FAKE_LIB=_fake_lib(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "run_bench_fake.py"), "--steps", "1", "--warmup", "0",
                          "--size", "16", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["dtype"] == "f32" and rec["data"] == "synthetic" and rec["vs_baseline"] is None
    # asking for more GPUs than the launcher provided is an error, not a silent single-GPU run
    bad = subprocess.run([sys.executable, os.path.join(HERE, "run_bench_fake.py"), "--gpus", "2", "--size", "16"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0
