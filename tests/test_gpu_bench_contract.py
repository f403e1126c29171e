"""The driver's contract for bench.py, checked on the GPU box with a very short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--cpu-seconds", "3", "--no-secondary"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "flow_err"):
        assert k in d, k
    # the metric's second half (BASELINE.json: "+ max-abs flow err vs TF CPU"): the last timed step's five flows, the output-resolution
    # flow and the warped frame against the CPU restatement run on the SAME first samples of the timed batch, within 1e-3
    fe, cb = d["flow_err"], d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and "timed GPU batch" in cb["sample"]
    assert fe["tol"] == 1e-3 and fe["samples"] == 2 and "parity unpinned" in fe["vs"]
    assert set(fe["max_abs"]) == {"predict_flow6", "predict_flow5", "predict_flow4", "predict_flow3", "predict_flow2", "outflow"}
    assert all(0 <= v <= 1e-3 for v in fe["max_abs"].values()), fe
    assert 0 <= fe["warped_max_abs_masked"] <= 1e-3 and fe["within_tol"] is True and fe["worst"] <= 1e-3
    assert fe["vs_fp64"]["within_tol"] is True and all(v <= 1e-3 for v in fe["vs_fp64"]["max_abs"].values())
    assert fe["max_abs_flow"]["predict_flow2"] > 0.5         # a real flow field, not zeros
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 8 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.3 < rf["frac"] < 1.0 and rf["kernel"].startswith("conv_") and "traffic_source" in rf
    assert d["metric"].endswith("@512x512") and "roofline_hbm" in d
    rh = d["roofline_hbm"]                                   # the glue + warp launch against the HBM peak
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["peak"] == 8000.0 and abs(rh["frac"] - rh["achieved"] / rh["peak"]) < 1e-3
    # the step's last launch: predict_flow2's gather + the flow glue + tf_warp (flow_ops.hip, pf2_glue_warp_kernel).  Algorithmic bytes per
    # output pixel at 512x512: 128 B per tap-table row (8.0) + the coarser flow (0.125) + predict_flow2 written once (7.94) + frame read,
    # output-resolution flow and warped frame written (32)
    assert 0.1 < rh["frac"] < 1.0 and rh["entry_point"].startswith("vstab_stabilise_originalsize") and rh["launches"] >= 1
    assert rh["kernel"].startswith("pf2_glue_warp_kernel") and abs(rh["alg_bytes_per_output_pixel"] - 48.06) < 0.2
    # the spatial-transformer / warp.py samplers ride along as rows of the same block (24 algorithmic bytes per output pixel)
    st_rows = [r_ for r_ in rh["other_rows"] if r_["row"].startswith(("S2 ", "S3 "))]
    assert len(st_rows) == 3 and all(abs(r_["alg_bytes_per_output_pixel"] - 24.0) < 1e-6 and 0.05 < r_["frac"] < 1.0 for r_ in st_rows)
    assert d["config"]["host_calls_per_step"].startswith("1 ")


def test_bench_self_launched_rccl_rank_prints_exactly_one_line():
    """The N > 1 path with one rank: the parent starts a torch.distributed.run child (no port picked ahead of time), the rank
    initialises RCCL -- which prints its version banner to file descriptor 1 on these boxes -- and all-gathers its uint8 frames;
    stdout must still be exactly the one JSON line."""
    env = dict(os.environ, VSTAB_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "self-launch:" in r.stderr and "--rdzv-endpoint=127.0.0.1:0" in r.stderr and "--master-port" not in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "uint8 warped frames" in d["config"]["all_gather"] and "over RCCL" in d["config"]["all_gather"]
    # SURVEY.md 8e "fps + gather time separately": the same steps again without the reassembly, after the timed region
    ag = d["all_gather"]
    for k in ("schedule", "transport", "bytes_per_rank_per_step", "bytes_gathered_per_step", "ms_per_step_with", "ms_per_step_without", "exposed_ms"):
        assert k in ag, k
    assert ag["transport"] == "RCCL" and ag["bytes_per_rank_per_step"] == 8 * 512 * 512 * 3 and ag["ms_per_step_with"] == d["ms_per_step"]
    assert abs(ag["exposed_ms"] - (ag["ms_per_step_with"] - ag["ms_per_step_without"])) < 1e-3 and ag["ms_per_step_without"] > 0


def test_bench_self_launches_two_ranks_rehearsal():
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks as a child job, they run the HIP path side by
    side (sharing this box's one GPU, frames reassembled through gloo / host memory: --backend gloo is a control-flow rehearsal,
    its throughput means nothing) and rank 0 alone prints the line, with the whole job's aggregate."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--batch", "2", "--height", "128", "--width", "128"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "self-launch:" in r.stderr and "--nproc-per-node=2" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3          # both ranks' samples / max-over-ranks time
    assert "schedule allgather" in d["config"]["all_gather"] and "gloo" in d["config"]["all_gather"] and "RCCL" not in d["config"]["all_gather"]
    ag = d["all_gather"]
    assert ag["transport"].startswith("gloo") and ag["bytes_gathered_per_step"] == 2 * 2 * 128 * 128 * 3 and ag["ms_per_step_without"] > 0


def test_bench_clip_two_ranks_ragged_rehearsal():
    """BASELINE configs[3] control flow with two self-launched ranks sharing this box's GPU (gloo, host-memory reassembly): 9 frames ->
    shards of 5 and 4, micro-batches of 2, the common part gathered per micro-batch, rank 0's fifth frame through finish()."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_clip.py"), "--gpus", "2", "--backend", "gloo", "--frames", "9",
                        "--height", "128", "--width", "160", "--micro-batch", "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["gathered_bytes"] == 9 * 128 * 160 * 3


@pytest.mark.parametrize("ranks,frames,mb", [(2, 11, 4), (3, 13, 4)])
def test_sharded_clip_is_bit_equal_to_the_unsharded_one(ranks, frames, mb):
    """SURVEY.md 4.4's multi-GPU test: an N-way sharded run, ragged tails included, reassembled over the process group, equals the
    one-rank run bit for bit.  Self-launched gloo ranks sharing this box's GPU (the HIP kernels compute, host memory reassembles):
    11 frames on 2 ranks = shards of 6 and 5 in micro-batches of 4 + 2 / 4 + 1; 13 frames on 3 ranks = 5 + 4 + 4.  The launch plan is
    pinned to the micro-batch (vstab_set_plan_batch), so a frame's bits do not depend on its micro-batch's size; bench_clip.py --check
    computes the unsharded sequence on rank 0 (micro-batches of 4 from frame 0) and compares."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_clip.py"), "--gpus", str(ranks), "--backend", "gloo", "--frames", str(frames),
                        "--height", "256", "--width", "256", "--micro-batch", str(mb), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == ranks and d["plan_batch"] == mb and d["sharded_equals_unsharded"] is True
