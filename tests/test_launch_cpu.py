"""`bench.py --gpus N` / `bench_clip.py --gpus N` start their own ranks: the spawned command,
its environment, and one real 2-rank child job on the CPU (gloo) whose stdout and exit code come back."""
import importlib.util
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCH_PY = os.path.join(ROOT, "coupe", "optical_flow_based_deep_video_stabilization_amd", "launch.py")


def _launch():
    spec = importlib.util.spec_from_file_location("vstab_launch_t", LAUNCH_PY)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_module_needs_no_package_import():
    # the parent of a self-launched job must not load the HIP library: launch.py imports only the stdlib
    src = open(LAUNCH_PY).read()
    assert "import torch" not in src and "_lib" not in src and "from ." not in src


def test_launch_command_and_env():
    L = _launch()
    env_in = {"PATH": "/usr/bin", "RANK": "3", "WORLD_SIZE": "9", "LOCAL_RANK": "3", "MASTER_PORT": "1", "MASTER_ADDR": "elsewhere"}
    cmd, env = L.launch_command("/x/bench.py", ["--gpus", "4", "--steps", "7"], 4, env=env_in, run_id="abc")
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    # the parent picks no port: the rendezvous store binds port 0 itself and the ranks re-use it
    assert "--master-port" not in cmd and "--rdzv-endpoint=127.0.0.1:0" in cmd and "--rdzv-backend=c10d" in cmd and "--rdzv-id=abc" in cmd
    assert cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"
    assert 1 <= int(env["OMP_NUM_THREADS"]) <= L.RANK_THREADS_CAP
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "7"]
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        assert k not in env                     # a stale rendezvous must not leak into the child job
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["VSTAB_SELF_LAUNCHED"] == "1"
    assert env["TORCHELASTIC_USE_AGENT_STORE"] == "True"                 # also when the caller's environment says otherwise
    assert L.launch_command("x.py", [], 2, env={"TORCHELASTIC_USE_AGENT_STORE": "0"})[1]["TORCHELASTIC_USE_AGENT_STORE"] == "True"
    with pytest.raises(ValueError):
        L.launch_command("/x/bench.py", [], 0)


def test_under_launcher_and_passthrough():
    L = _launch()
    assert L.under_launcher({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert not L.under_launcher({"RANK": "0"})
    clean = {k: v for k, v in os.environ.items() if k not in L.LAUNCHER_ENV}
    old = dict(os.environ)
    try:
        os.environ.clear(); os.environ.update(clean)
        assert L.maybe_self_launch("/x/none.py", [], 1) is None          # plain single-process run: carry on
        os.environ.update({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
        assert L.maybe_self_launch("/x/none.py", [], 2) is None          # already a rank: carry on
    finally:
        os.environ.clear(); os.environ.update(old)


def test_bench_scripts_self_launch_before_gpu_init():
    for name in ("bench.py", "bench_clip.py"):
        src = open(os.path.join(ROOT, name)).read()
        i_launch, i_gpu = src.index("maybe_self_launch("), src.index("torch.cuda.is_available()")
        assert i_launch < i_gpu, name
        assert "launch with torch.distributed.run" not in src, name


def test_real_two_rank_child_job_relays_stdout_and_rc(tmp_path):
    L = _launch()
    script = tmp_path / "mini.py"
    script.write_text(textwrap.dedent(f"""
        import importlib.util, json, os, sys
        spec = importlib.util.spec_from_file_location("l", {LAUNCH_PY!r}); L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
        rc = L.maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], int(sys.argv[1]))
        if rc is not None:
            raise SystemExit(rc)
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(dist.get_rank() + 1)]); dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({{"world": dist.get_world_size(), "sum": float(t), "self": os.environ.get("VSTAB_SELF_LAUNCHED")}}), flush=True)
        dist.destroy_process_group()
        raise SystemExit(int(sys.argv[2]))
    """))
    env = {k: v for k, v in os.environ.items() if k not in L.LAUNCHER_ENV}
    r = subprocess.run([sys.executable, str(script), "2", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"world": 2, "sum": 3.0, "self": "1"}
    r = subprocess.run([sys.executable, str(script), "2", "5"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0                       # a failing rank fails the parent


def test_rank_threads_come_from_the_cpu_share_not_cpu_count():
    L = _launch()
    share = L.cpu_share()
    assert 1 <= share <= (os.cpu_count() or 1)
    try:
        assert share <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    assert L.rank_threads(1, share=16) == L.RANK_THREADS_CAP and L.rank_threads(8, share=16) == 2 and L.rank_threads(8, share=4) == 1
    # a caller's own OMP_NUM_THREADS wins
    _, env = L.launch_command("/x/b.py", [], 2, env={"OMP_NUM_THREADS": "3"})
    assert env["OMP_NUM_THREADS"] == "3"


def test_rendezvous_failure_is_retried_once_and_only_then(monkeypatch):
    L = _launch()
    assert L.rendezvous_failure(1, "torch.distributed.DistNetworkError: ... port: 34911 ... EADDRINUSE", False)
    assert not L.rendezvous_failure(1, "EADDRINUSE", True)             # a rank already printed: not a rendezvous failure
    assert not L.rendezvous_failure(0, "EADDRINUSE", False)
    assert not L.rendezvous_failure(1, "RuntimeError: HIP error", False)
    calls = []

    def fake(cmd, env, tail_bytes=0):
        calls.append(cmd)
        return (1, "DistNetworkError ... EADDRINUSE", False) if len(calls) == 1 else (0, "", True)

    monkeypatch.setattr(L, "_run_child", fake)
    clean = {k: v for k, v in os.environ.items() if k not in L.LAUNCHER_ENV}
    monkeypatch.setattr(os, "environ", clean)
    assert L.maybe_self_launch("/x/b.py", ["--gpus", "2"], 2) == 0 and len(calls) == 2
    assert calls[0][-3:] == calls[1][-3:] and calls[0] != calls[1]                 # a fresh rendezvous id
    calls.clear()
    monkeypatch.setattr(L, "_run_child", lambda cmd, env, tail_bytes=0: (calls.append(cmd), (7, "DistNetworkError", False))[1])
    assert L.maybe_self_launch("/x/b.py", ["--gpus", "2"], 2) == 7 and len(calls) == 2      # twice, not forever
    calls.clear()
    monkeypatch.setattr(L, "_run_child", lambda cmd, env, tail_bytes=0: (calls.append(cmd), (3, "a rank failed", True))[1])
    assert L.maybe_self_launch("/x/b.py", ["--gpus", "2"], 2) == 3 and len(calls) == 1      # an ordinary failure is not retried


MINI_SLEEPER = """
import importlib.util, json, os, sys, time
spec = importlib.util.spec_from_file_location("l", {launch!r}); L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
rc = L.maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], 2)
if rc is not None:
    raise SystemExit(rc)
import torch, torch.distributed as dist
dist.init_process_group("gloo")
open(os.path.join({dir!r}, "pid%s" % os.environ["RANK"]), "w").write(str(os.getpid()))
dist.barrier()
if sys.argv[1] == "sleep":
    time.sleep(120)
if dist.get_rank() == 0:
    print(json.dumps({{"ok": True}}), flush=True)
dist.destroy_process_group()
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:            # a zombie still answers kill(0)
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_two_jobs_at_once_and_a_taken_default_port(tmp_path):
    # 29500 (torchrun's default master port) is taken and two self-launched jobs start at the same moment: no port is ever
    # chosen ahead of binding it, so both rendezvous
    import socket
    L = _launch()
    taken = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        taken.bind(("127.0.0.1", 29500))
        taken.listen(1)
    except OSError:
        pass                                         # somebody else holds it: just as good
    env = {k: v for k, v in os.environ.items() if k not in L.LAUNCHER_ENV}
    procs = []
    for i in range(2):
        d = tmp_path / f"job{i}"
        d.mkdir()
        script = d / "mini.py"
        script.write_text(MINI_SLEEPER.format(launch=LAUNCH_PY, dir=str(d)))
        procs.append(subprocess.Popen([sys.executable, str(script), "run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    try:
        for p in procs:
            out, err = p.communicate(timeout=300)
            assert p.returncode == 0, err[-2000:]
            assert [l for l in out.splitlines() if l.startswith("{")] == ['{"ok": true}']
    finally:
        taken.close()


@pytest.mark.parametrize("signame", ["SIGTERM", "SIGKILL"])
def test_signal_to_the_parent_leaves_no_rank_behind(tmp_path, signame):
    # SIGTERM is forwarded to the child job's process group; SIGKILL cannot be handled -- the agent then gets SIGTERM through
    # PR_SET_PDEATHSIG and takes its ranks down (what `timeout -k` around a bench relies on)
    import signal
    import time
    L = _launch()
    script = tmp_path / "mini.py"
    script.write_text(MINI_SLEEPER.format(launch=LAUNCH_PY, dir=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in L.LAUNCHER_ENV}
    p = subprocess.Popen([sys.executable, str(script), "sleep"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    try:
        t_end = time.time() + 240
        while time.time() < t_end and not all((tmp_path / f"pid{r}").exists() and (tmp_path / f"pid{r}").read_text() for r in (0, 1)):
            assert p.poll() is None, p.stderr.read()[-2000:]
            time.sleep(0.2)
        pids = [int((tmp_path / f"pid{r}").read_text()) for r in (0, 1)]
        assert all(_alive(x) for x in pids)
        p.send_signal(getattr(signal, signame))
        rc = p.wait(timeout=60)
        assert rc != 0
        t_end = time.time() + 30
        while time.time() < t_end and any(_alive(x) for x in pids):
            time.sleep(0.2)
        assert not any(_alive(x) for x in pids), "a rank survived its parent's " + signame
    finally:
        if p.poll() is None:
            p.kill()


def test_claim_stdout_keeps_library_chatter_off_the_json_line(tmp_path):
    # RCCL prints its version banner to stdout (file descriptor 1) on the GPU boxes; after claim_stdout() only what is written to
    # the returned file reaches the real stdout
    script = tmp_path / "chatty.py"
    script.write_text(textwrap.dedent(f"""
        import importlib.util, os, sys
        spec = importlib.util.spec_from_file_location("l", {LAUNCH_PY!r}); L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
        real = L.claim_stdout()
        os.write(1, b"RCCL version : 2.26.6\\n")          # a C library writing to fd 1
        print("python chatter")
        print('{{"value": 1}}', file=real, flush=True)
    """))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout == '{"value": 1}\n'
    assert "RCCL version" in r.stderr and "python chatter" in r.stderr
