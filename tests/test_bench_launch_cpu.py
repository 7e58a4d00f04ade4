"""`python bench.py --gpus N` starts its own ranks: the self-launch path, the rendezvous and the timed-region protocol
on CPU over gloo with a stub step (EDV_BENCH_STUB=1).  The explicit torch.distributed.run form the driver uses stays covered as well."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ, EDV_BENCH_STUB="1", EDV_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus_2_starts_its_own_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"], env=_env(), capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line["stub"] and line["n_gpus"] == 2 and line["steps"] == 4 and line["ms_per_step"] >= 5.0


def test_bench_under_an_explicit_torchrun_still_works():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                        str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(), capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 2


def test_a_failing_rank_fails_the_launcher():
    env = _env()
    env["EDV_BENCH_BACKEND"] = "no-such-backend"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=240)
    assert r.returncode != 0
