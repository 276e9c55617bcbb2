"""bench.py's driver contract: one JSON line with the agreed fields, the roofline object of the dominant kernel, the CPU
baseline and the parity figures; also under torch.distributed.run with one rank (the RCCL init / barrier path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cmd, env=None):
    out = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]    # exactly ONE line on stdout, and it is the JSON line
    return json.loads(lines[0])                                               # (libraries' banners -- RCCL's -- go to stderr)


def test_bench_line_has_the_contract_fields():
    d = _line([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--cpu-scenes", "1"])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["unit"] == "scenes/s"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["batch_per_gpu"] == 8
    assert abs(d["value"] - 8 * 6 / (d["ms_per_step"] * 6 / 1e3)) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    # utilisation figures count EXECUTED flops; the algorithmic count is kept beside them
    assert r["frac"] == r["frac_executed"] <= r["frac_algorithmic"] and 0.3 < r["step_frac_executed"] < 1
    assert r["executed_flop_per_launch"] <= r["algorithmic_units_per_launch"]
    if r["rocprof_avg_launch_ms"] is not None:
        assert r["rocprof_source"].startswith("profiles/") and 0 < r["frac_rocprof"] < 1
    assert d["config"]["executed_gflop_per_scene"] <= d["config"]["launched_gflop_per_scene"] <= d["config"]["scorenet_gflop_per_scene"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    p = d["parity"]
    assert p["region_indices_equal"] is True and p["score_max_abs_err"] <= p["tolerance"] == 1e-4
    assert p["grasp_max_abs_err"] <= 1e-4 and p["feature_max_rel_err"] <= 1e-4


def test_bench_under_torch_distributed_run_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "4", "--warmup", "1", "--cpu-scenes", "1"], env)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and "cpu_baseline" in d


def test_bench_line_carries_the_round3_fields():
    d = _line([sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--cpu-scenes", "0", "--latency-runs", "0",
               "--no-lookahead-steps", "4", "--train-steps", "6"])
    assert d["config"]["sampling_lookahead_batches"] == 6          # all six batches of the run were sampled by the first launch
    assert d["value_no_lookahead"] > 0 and "fps-group 1" in d["value_no_lookahead_note"]
    assert d["config"]["hip_graphs"] is False                       # 8 x 25 600 is not a launch-bound shape: one launch per kernel
    t = d["train"]
    assert t["scenes_per_s"] > 0 and t["ms_per_step"] > 0 and t["roofline"]["kernel"] == "tgemm_kernel"
    assert t["allreduce_ms"] is None                                # one rank: no collective
    assert 0 < t["host_ms_per_step"] <= t["ms_per_step"] * 1.05 and t["host_blocked_ms_per_step"] >= 0
    assert t["distinct_batches"] == 4 and t["bucket_bytes"] is None
    # every timed iteration by itself (HIP events), its median, the launching thread's time per iteration, the blocking reads;
    # all six iterations replayed hipGraphs (train_step._TrunkGraphs, captured during the five warm-up iterations), and the
    # trunk stream's phases add up to an iteration
    assert len(t["iteration_ms"]) == len(t["host_iteration_ms"]) == 6 and t["graph_replays"] == 6
    assert min(t["iteration_ms"]) <= t["ms_per_step_median"] <= max(t["iteration_ms"])
    ph = t["trunk_stream"]
    assert ph["iterations"] == 6 and 0 < ph["busy_ms"] <= t["ms_per_step_median"] * 1.02
    assert abs(ph["busy_ms"] + ph["region_wait_ms"] - t["ms_per_step_median"]) <= 0.1 * t["ms_per_step_median"]
    assert t["readback_wait_ms"] and all(r["median_ms"] >= 0 for r in t["readback_wait_ms"])
    assert "hipGraphs" in t["roofline"]["accounting_note"]
    g = t["gpu_timeline"]
    assert g is not None and ("error" in g or (g["busy_ms_per_step"] > 0 and g["idle_ms_per_step"] >= 0))


def test_bench_small_batch_replays_graphs_and_says_so():
    """One scene per batch: the geometry + feature stages are hipGraph replays (pipeline._StageGraphs); the line says so and
    takes its per-kernel accounting from extra launch-by-launch steps outside the timed region."""
    d = _line([sys.executable, "bench.py", "--batch", "1", "--steps", "12", "--warmup", "3", "--cpu-scenes", "1", "--latency-runs", "0",
               "--no-lookahead-steps", "0", "--train-steps", "0"])
    assert d["config"]["hip_graphs"] is True and d["config"]["batch_per_gpu"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert "hipGraphs" in r["accounting_note"] and 0 < r["frac"] < 1 and r["kernel"]
    p = d["parity"]
    assert p["region_indices_equal"] is True and p["score_max_abs_err"] <= 1e-4


def test_bench_two_ranks_over_gloo_on_one_device():
    """The multi-rank paths of bench.py -- barrier + max-over-ranks around the timed region, scene sharding by rank, the
    train object's gradient all-reduce through GradientBucket -- driven with two ranks on the ONE GPU of the test box
    (REGNET_BENCH_ONE_DEVICE_GLOO: gloo instead of RCCL, which refuses two ranks on one device).  Small scenes, few steps:
    this checks that the line comes out and adds up, not its speed."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", REGNET_BENCH_ONE_DEVICE_GLOO="1")
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29544", "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--batch", "2", "--points", "6144", "--train-steps", "2", "--train-batch", "1"], env)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak"
    assert abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-2 * d["value"]     # whole-job scenes / max-over-ranks time
    assert "cpu_baseline" not in d and d["value_no_lookahead"] is None                    # single-rank extras stay off
    t = d["train"]
    assert "error" not in t, t
    assert t["allreduce_ms"] is not None and t["allreduce_ms"] >= 0 and t["parallelism"].startswith("dp2: ONE flat")
    # the line describes the process group it ran on: backend / world size as the group reports them, the
    # rank-id all-reduce proving two distinct ranks, the bucket-sized probe all-reduce
    c = d["config"]["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["distinct_ranks_by_allreduce"] == 2
    assert c["probe_bytes"] == (7066927 + 128) * 4 and c["probe_allreduce_ms"] > 0 and c["bus_GBps"] > 0
    assert c["rccl_version"] is None                                # gloo here; under RCCL the version tuple is reported
    assert 28e6 < t["bucket_bytes"] < 29e6          # both networks' 7 066 927 gradients + one presence flag per parameter, fp32
