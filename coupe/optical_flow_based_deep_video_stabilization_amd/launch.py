"""Self-launch of the one-process-per-GPU benches.

`python bench.py --gpus N` must work without an external launcher (the reference is one process with one
tf.Session, main:489; its user types one command).  A process that is NOT already a rank of a launcher
starts `python -m torch.distributed.run --nproc-per-node N <script> <same arguments>` as a CHILD, relays
its stdout/stderr/exit code and does nothing else -- in particular it never initialises the GPU (on this
pool a process that has touched the GPU must not exec or be replaced; the parent only waits).

What makes the path boring:
* the parent picks NO port.  The child job's rendezvous store binds 127.0.0.1:0 (the kernel assigns the port
  while binding) and the ranks re-use that store (TORCHELASTIC_USE_AGENT_STORE), so there is no window between
  choosing a port and listening on it (a bind-then-close pick once lost that race: EADDRINUSE in torchrun's
  rendezvous, round 2);
* a child that dies in the rendezvous before any rank ran is started again ONCE (a new process; the parent has
  no GPU state to carry over);
* the child job runs in its own session; SIGTERM / SIGINT / SIGHUP to the parent are forwarded to the whole
  group, SIGKILL follows after a grace period, and the agent asks for SIGTERM should the parent itself be
  killed (PR_SET_PDEATHSIG) -- `timeout -k` around the bench leaves no rank holding the GPU;
* OMP_NUM_THREADS of the ranks comes from the CPUs this job may really use (affinity mask and cgroup quota),
  not from os.cpu_count(): on a 16-CPU share of a 256-thread host the latter made every OpenMP region of a
  rank spin 256 threads on 16 CPUs, and the Python thread that issues the kernels got a sixteenth of a core
  (5-6 ms per 3 ms step, round 2's "host-bound self-launched runs").
"""
from __future__ import annotations

import os
import signal
import subprocess
import sys
import threading
import time
import uuid
from typing import Dict, List, Optional, Tuple

LAUNCHER_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK")
RANK_THREADS_CAP = 8            # a rank's host side is one Python thread issuing kernels; the CPU baseline sets its own count
RENDEZVOUS_ERRORS = ("EADDRINUSE", "address already in use", "DistNetworkError", "RendezvousConnectionError",
                     "RendezvousTimeoutError", "The server socket has failed to listen")
GRACE_S = 10.0


def under_launcher(env=None) -> bool:
    """True when this process is already a rank started by torch.distributed.run / torchrun."""
    env = os.environ if env is None else env
    return all(k in env for k in LAUNCHER_ENV)


def cpu_share() -> int:
    """CPUs this process tree may really use: the affinity mask, cut by a cgroup CPU quota if one is set."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, n)


def rank_threads(nproc: int, share: Optional[int] = None) -> int:
    share = cpu_share() if share is None else share
    return max(1, min(RANK_THREADS_CAP, share // max(1, nproc)))


def launch_command(script: str, argv: List[str], nproc: int, env=None, run_id: Optional[str] = None) -> Tuple[List[str], Dict[str, str]]:
    """The child command line and environment for `nproc` ranks of `script argv` on this node."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    env = dict(os.environ if env is None else env)
    for k in LAUNCHER_ENV + ("MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(rank_threads(nproc)))
    env["VSTAB_SELF_LAUNCHED"] = "1"
    # the ranks re-use the agent's rendezvous store (which bound 127.0.0.1:0) instead of a MASTER_PORT the agent would pick by
    # bind-then-close: set it, whatever the environment or the installed torch's default says
    env["TORCHELASTIC_USE_AGENT_STORE"] = "True"
    run_id = run_id or uuid.uuid4().hex[:12]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id={run_id}", "--local-addr", "127.0.0.1",
           os.path.abspath(script)] + list(argv)
    return cmd, env


def _resolve_prctl():
    """libc's prctl, looked up in the PARENT: between fork and exec of a process that has threads (torch is imported) only the call
    itself is made -- no import, no dlopen."""
    try:
        import ctypes
        return ctypes.CDLL(None, use_errno=True).prctl
    except Exception:
        return None


_PRCTL = _resolve_prctl()


def _ask_for_sigterm_when_parent_dies():          # runs in the child between fork and exec
    if _PRCTL is not None:
        try:
            _PRCTL(1, int(signal.SIGTERM), 0, 0, 0)      # PR_SET_PDEATHSIG
        except Exception:
            pass


def _kill_group(proc: subprocess.Popen, sig: int):
    try:
        os.killpg(proc.pid, sig)                 # start_new_session: the child's pid is its process-group id
    except (ProcessLookupError, PermissionError):
        pass


def _run_child(cmd: List[str], env: Dict[str, str], tail_bytes: int = 16384) -> Tuple[int, str, bool]:
    """Runs one child job; returns (exit code, tail of its stderr, whether it wrote anything to stdout).
    Both streams are relayed byte for byte as they arrive (the child's JSON line is ours); the relay is what
    tells a rendezvous failure -- nothing on stdout, a rendezvous error on stderr -- from a failing rank."""
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True,
                            preexec_fn=_ask_for_sigterm_when_parent_dies)
    state = {"tail": b"", "wrote_stdout": False, "signalled": None}

    def relay(src, dst, is_err):
        while True:
            chunk = src.read1(65536) if hasattr(src, "read1") else src.read(65536)
            if not chunk:
                break
            if is_err:
                state["tail"] = (state["tail"] + chunk)[-tail_bytes:]
            else:
                state["wrote_stdout"] = True
            try:
                dst.write(chunk); dst.flush()
            except (BrokenPipeError, ValueError):
                pass

    threads = [threading.Thread(target=relay, args=(proc.stdout, sys.stdout.buffer, False), daemon=True),
               threading.Thread(target=relay, args=(proc.stderr, sys.stderr.buffer, True), daemon=True)]
    for t in threads:
        t.start()

    def forward(signum, _frame):
        state["signalled"] = signum
        _kill_group(proc, signum)

    old = {}
    if threading.current_thread() is threading.main_thread():
        for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
            old[s] = signal.signal(s, forward)
    try:
        deadline = None
        while True:
            try:
                rc = proc.wait(timeout=0.2)
                break
            except subprocess.TimeoutExpired:
                pass
            if state["signalled"] is not None:
                if deadline is None:
                    deadline = time.monotonic() + GRACE_S
                elif time.monotonic() > deadline:
                    _kill_group(proc, signal.SIGKILL)
                    deadline = float("inf")
        if state["signalled"] is not None:
            # the agent is gone; make sure no rank of its group outlives it
            t_end = time.monotonic() + GRACE_S
            while time.monotonic() < t_end:
                try:
                    os.killpg(proc.pid, 0)
                except (ProcessLookupError, PermissionError):
                    break
                time.sleep(0.1)
            else:
                _kill_group(proc, signal.SIGKILL)
    finally:
        for s, h in old.items():
            signal.signal(s, h)
    for t in threads:
        t.join(timeout=5)
    if state["signalled"] is not None and rc == 0:
        rc = 128 + int(state["signalled"])
    return rc, state["tail"].decode("utf-8", "replace"), state["wrote_stdout"]


def rendezvous_failure(rc: int, stderr_tail: str, wrote_stdout: bool) -> bool:
    """A failed child whose ranks never reported and whose stderr shows the job died while rendezvousing."""
    return rc != 0 and not wrote_stdout and any(m in stderr_tail for m in RENDEZVOUS_ERRORS)


def maybe_self_launch(script: str, argv: List[str], nproc: int, force: bool = False) -> Optional[int]:
    """If `nproc` ranks are wanted and this process is not one of them, run them as a child job and return
    its exit code (the caller exits with it).  Returns None when the caller should carry on as a rank
    (already under a launcher, or a plain single-process run).  Must be called before any GPU call."""
    if under_launcher():
        return None
    if nproc <= 1 and not force:
        return None
    cmd, env = launch_command(script, argv, max(nproc, 1))
    print("self-launch: " + " ".join(cmd) + f"   (OMP_NUM_THREADS={env['OMP_NUM_THREADS']}, cpu share {cpu_share()})", file=sys.stderr, flush=True)
    rc, tail, wrote = _run_child(cmd, env)
    if rendezvous_failure(rc, tail, wrote):
        print("self-launch: the child job failed in its rendezvous before any rank ran; starting ONE fresh child", file=sys.stderr, flush=True)
        cmd, env = launch_command(script, argv, max(nproc, 1))
        rc, tail, wrote = _run_child(cmd, env)
    return rc


def claim_stdout():
    """The bench contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to stdout when
    NCCL_DEBUG is VERSION or higher, as it is on the GPU boxes): point file descriptor 1 at stderr for the rest of the process
    and return a file object on the ORIGINAL stdout for the one line that belongs there.  Call before any library initialises."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real
