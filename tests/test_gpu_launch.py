"""The real multi-process path of `bench.py --gpus N`, executed before the day it counts: two ranks started by
the script's own launcher, both on device 0 (a one-GPU lease is all there is), the all-to-all staged through host memory under gloo
because RCCL refuses two ranks on one device.  Everything else is the path the 8-GPU job takes: the sub-slab cut, the halos, the
round-robin pipeline, each rank's receiver over its channel shard, the verification on every rank, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("world", [2, 4])
def test_bench_two_ranks_rehearsed_on_one_gpu(world):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1",
           "--reps", "2", "--no-cpu", "--frames", "4", "--slabs", "2", "--serial-steps", "1"]
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])          # (a rank whose verification fails exits non-zero)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                # one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["verified"]["ok"] and d["verified"]["frames"] == d["verified"]["expected"] > 0
    assert d["config"]["rehearsal_on_one_gpu"] is True and "sharding.Pipeline" in d["config"]["multi_gpu_path"]
    assert d["steps"] == 2 and d["repetitions"] == 2 and d["value"] > 0 and "exchange" in d
    assert d["config"]["receiver_hints_from_the_benchmark"].startswith("none")
    # the fields the first real SCALE record will be read by: which exchange ran, how many ranks the transport counted,
    # how many ranks a real all_reduce summed over -- in the rehearsal the exchange is NOT RCCL and the line has to say so
    x = d["exchange"]
    assert x["ranks_seen_by_all_reduce"] == world and x["path"].startswith("rehearsal") and "NOT RCCL" in x["path"] and x["rccl_ranks"] is None


def test_a_hung_rendezvous_still_prints_a_line():
    """Two ranks are announced and one is started: the rendezvous can never complete.  bench.py's watchdog makes rank 0 print a line with
    value 0 and the stage that hung, and exit non-zero, instead of sitting there until the driver's clock runs out."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(clean_env(), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "1", "--warmup", "0", "--no-cpu",
           "--contact-timeout", "8"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] == 0.0 and "timed out" in d["error"] and "rendezvous" in d["error"] and d["n_gpus"] == 2


def test_c_pipeline_failure_falls_back_on_every_rank():
    """`--exchange auto` takes the C-ABI pipeline (grouped ncclSend / ncclRecv) for N > 1 -- code that has never run between two GPUs.
    bench.py therefore creates it and runs one trial step under a guard, lets the ranks agree on the outcome, and takes the torch
    pipeline on a fresh receiver everywhere if any rank failed.  Two ranks on ONE device make RCCL refuse the communicator
    (BENCH_REHEARSE_C=1 lets the rehearsal try it): the fall-back has to carry the run, with the reason in the line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--steps", "2", "--warmup", "1",
           "--reps", "1", "--no-cpu", "--frames", "4", "--slabs", "2", "--serial-steps", "1"]
    r = subprocess.run(cmd, env=dict(clean_env(), BENCH_REHEARSE_C="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["verified"]["ok"] and d["verified"]["frames"] == d["verified"]["expected"] > 0
    path = d["config"]["multi_gpu_path"]
    assert "sharding.Pipeline" in path and "C-ABI pipeline failed" in path, path
