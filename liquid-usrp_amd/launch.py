"""One process per GPU without an external launcher.

`python bench.py --gpus 8` has to be enough: when a multi-rank job is asked for and the process was not started by
torch.distributed.run (no WORLD_SIZE in the environment), ensure_ranks() re-executes the same command line under

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>

and exits with its status; under torch.distributed.run (the driver's form) it returns the rank triple.  The rendezvous
address is always 127.0.0.1 (the container host name may not resolve).
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ranks_from_env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0")))


def launch_command(gpus, script, argv, port=None):
    """The torch.distributed.run command line that starts `gpus` ranks of `script argv...` on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % gpus,
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def ensure_ranks(gpus, script=None, argv=None):
    """-> (rank, world, local_rank).  gpus > 1 outside a launcher: start the ranks and exit with their status."""
    if gpus < 1:
        sys.exit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if gpus == 1:
            return 0, 1, 0
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this driver
        env.setdefault("MASTER_ADDR", "127.0.0.1")
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = launch_command(gpus, script or os.path.abspath(sys.argv[0]), sys.argv[1:] if argv is None else argv)
        sys.stdout.flush(); sys.stderr.flush()
        sys.exit(subprocess.call(cmd, env=env))
    rank, world, local = ranks_from_env()
    if world != gpus:
        sys.exit("--gpus %d but the launcher started %d ranks (torch.distributed.run --nproc-per-node %d, or no launcher at all)"
                 % (gpus, world, gpus))
    return rank, world, local


def dry_run(gpus):
    """Launcher check that needs no GPU: the ranks rendezvous under gloo, all-reduce their rank numbers, rank 0 prints
    one JSON line.  (`bench.py --gpus N --dry-run-launch`; tests/test_launch.py)"""
    import json
    rank, world, local = ensure_ranks(gpus)
    total = rank
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        total = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launcher": "ok", "n_gpus": world, "rank_sum": total, "expected": world * (world - 1) // 2}))
    return 0 if total == world * (world - 1) // 2 else 1
