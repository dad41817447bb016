"""`python bench.py --gpus N` must be a complete command (missing #1): without a launcher around it the
script starts its own ranks under torch.distributed.run on 127.0.0.1; under torch.distributed.run (the driver's form)
it takes the ranks from the environment.  Checked here without a GPU: --dry-run-launch makes the ranks rendezvous
under gloo, all-reduce their rank numbers and print one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


@pytest.mark.parametrize("script", ["bench.py", "bench_duplex.py"])
def test_self_launch_two_ranks(script):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2", "--dry-run-launch"],
                       env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = last_json(r.stdout)
    assert out == {"launcher": "ok", "n_gpus": 2, "rank_sum": 1, "expected": 1}


def test_under_torchrun_keeps_working():
    sys.path.insert(0, os.path.join(ROOT, "liquid-usrp_amd"))
    import launch
    cmd = launch.launch_command(2, os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--dry-run-launch"])
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_gpus"] == 2


def test_single_rank_needs_no_launcher_and_mismatch_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run-launch"],
                       env=clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and last_json(r.stdout)["n_gpus"] == 1
    env = clean_env()
    env.update(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")               # a launcher that started 4 ranks for --gpus 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-launch"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "started 4 ranks" in r.stderr
