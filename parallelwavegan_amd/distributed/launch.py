#!/usr/bin/env python3
"""One-process-per-GPU launcher (command-line compatible with ``parallel_wavegan.distributed.launch``,
/root/reference/parallel_wavegan/distributed/launch.py:15-171, as used by
egs/*/voc1/run.sh stage 2: ``launch.py --nproc_per_node N -c parallel-wavegan-train ...``).

Differences that matter on an MI355X node:

* every rank gets ``LOCAL_RANK`` *and* (unless ``--use_env``) ``--local_rank=N`` like the reference;
* ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is set when the caller has not set it.  Evidence: the platform notes of this
  ROCm 7.2 image state that the host driver supports only dmabuf IPC and that RCCL / cross-process device-memory
  sharing otherwise fail with ``hipIpcGetMemHandle: invalid argument``; the image itself exports the variable, so
  this default only matters for environments built from scratch (``env -i``, service managers).  It is a
  ``setdefault``: an explicit value of the caller always wins;
* the ranks are supervised together: the first rank that fails takes the others down (the reference
  waits for the ranks one after the other, so a crashed rank 1 leaves rank 0 blocked in a collective
  forever), and the launcher's exit status is that rank's;
* ``--master_port 0`` picks a free port.

Python API: :func:`spawn` (used by ``bench.py --gpus N`` to launch itself).
"""
import os
import signal
import socket
import subprocess
import sys
import time
from argparse import REMAINDER, ArgumentParser


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank, local_rank, world_size, master_addr, master_port, nproc_per_node, base=None):
    env = dict(os.environ if base is None else base)
    env.update(MASTER_ADDR=str(master_addr), MASTER_PORT=str(master_port), WORLD_SIZE=str(world_size),
               RANK=str(rank), LOCAL_RANK=str(local_rank), LOCAL_WORLD_SIZE=str(nproc_per_node))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "OMP_NUM_THREADS" not in env and nproc_per_node > 1:
        # the host side of a rank is launch-bound, not compute-bound: leave the cores to the data loaders
        env["OMP_NUM_THREADS"] = str(max(1, (os.cpu_count() or 1) // (2 * nproc_per_node)))
    return env


def supervise(procs, poll_s=0.2):
    """Wait for all ranks; if one fails, terminate the rest.  Returns the first non-zero exit status (or 0)."""
    status = 0
    alive = list(procs)
    try:
        while alive:
            for p in list(alive):
                rc = p.poll()
                if rc is None:
                    continue
                alive.remove(p)
                if rc != 0 and status == 0:
                    status = rc
                    for q in alive:  # a collective with a dead peer never returns
                        q.terminate()
            if alive:
                time.sleep(poll_s)
    except KeyboardInterrupt:
        for q in alive:
            q.send_signal(signal.SIGINT)
        status = 130
    finally:
        deadline = time.time() + 10.0
        for q in alive:
            try:
                q.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
    return status


def spawn(cmd, nproc_per_node, nnodes=1, node_rank=0, master_addr="127.0.0.1", master_port=0, local_rank_at=None,
          env=None):
    """Start ``cmd`` (argv list) once per local rank with the rendezvous environment set; returns the
    exit status.  ``local_rank_at``: argv index at which ``--local_rank=N`` is inserted (the legacy
    convention of the reference: right after the script name); None = environment only."""
    port = int(master_port) or free_port()
    world = nproc_per_node * nnodes
    procs = []
    for local_rank in range(nproc_per_node):
        rank = nproc_per_node * node_rank + local_rank
        argv = list(cmd)
        if local_rank_at is not None:
            argv.insert(local_rank_at, f"--local_rank={local_rank}")
        procs.append(subprocess.Popen(argv, env=rank_env(rank, local_rank, world, master_addr, port, nproc_per_node, env)))
    return supervise(procs)


def parse_args(argv=None):
    parser = ArgumentParser(description="spawn one training process per MI355X of this node")
    parser.add_argument("--nnodes", type=int, default=1, help="number of nodes")
    parser.add_argument("--node_rank", type=int, default=0, help="rank of this node")
    parser.add_argument("--nproc_per_node", type=int, default=1, help="processes (= GPUs) on this node")
    parser.add_argument("--master_addr", default="127.0.0.1", type=str, help="address of rank 0")
    parser.add_argument("--master_port", default=29500, type=int, help="port of rank 0 (0 = pick a free one)")
    parser.add_argument("--use_env", default=False, action="store_true",
                        help="do not append --local_rank=N to the script arguments (LOCAL_RANK is always set)")
    parser.add_argument("-m", "--module", default=False, action="store_true", help="run the script as `python -m`")
    parser.add_argument("-c", "--command", default=False, action="store_true", help="the script is a command")
    parser.add_argument("training_script", type=str, help="script / module / command, followed by its arguments")
    parser.add_argument("training_script_args", nargs=REMAINDER)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.command:
        cmd = [args.training_script]
    else:
        cmd = [sys.executable, "-u"] + (["-m"] if args.module else []) + [args.training_script]
    head = len(cmd)
    cmd += args.training_script_args
    status = spawn(cmd, args.nproc_per_node, args.nnodes, args.node_rank, args.master_addr, args.master_port,
                   local_rank_at=None if args.use_env else head)
    if status != 0:
        sys.exit(status)


if __name__ == "__main__":
    main()
