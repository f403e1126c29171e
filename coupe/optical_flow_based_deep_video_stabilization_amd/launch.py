"""Self-launch of the one-process-per-GPU benches.

`python bench.py --gpus N` must work without an external launcher (the reference is one process with one
tf.Session, main:489; its user types one command).  A process that is NOT already a rank of a launcher
starts `python -m torch.distributed.run --nproc-per-node N <script> <same arguments>` as a CHILD, relays
its stdout/stderr/exit code and does nothing else -- in particular it never initialises the GPU (on this
pool a process that has touched the GPU must not exec or be replaced; the parent only waits).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import Dict, List, Optional, Tuple

LAUNCHER_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK")


def under_launcher(env=None) -> bool:
    """True when this process is already a rank started by torch.distributed.run / torchrun."""
    env = os.environ if env is None else env
    return all(k in env for k in LAUNCHER_ENV)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(script: str, argv: List[str], nproc: int, port: Optional[int] = None,
                   env=None) -> Tuple[List[str], Dict[str, str]]:
    """The child command line and environment for `nproc` ranks of `script argv` on this node."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    env = dict(os.environ if env is None else env)
    for k in LAUNCHER_ENV + ("MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or nproc) // nproc)))
    env["VSTAB_SELF_LAUNCHED"] = "1"
    port = free_port() if port is None else int(port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    return cmd, env


def maybe_self_launch(script: str, argv: List[str], nproc: int, force: bool = False) -> Optional[int]:
    """If `nproc` ranks are wanted and this process is not one of them, run them as a child job and return
    its exit code (the caller exits with it).  Returns None when the caller should carry on as a rank
    (already under a launcher, or a plain single-process run).  Must be called before any GPU call."""
    if under_launcher():
        return None
    if nproc <= 1 and not force:
        return None
    cmd, env = launch_command(script, argv, max(nproc, 1))
    print("self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)      # the child inherits stdout/stderr: its JSON line is ours
