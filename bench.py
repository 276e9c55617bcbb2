#!/usr/bin/env python
"""bench.py -- REGNet forward hot path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the forward hot path (ScoreNet -> region grouping -> grasp-region +
refine network, eval mode) over one batch of synthetic 25 600-point scenes that is already
resident in HBM.  Workload = BASELINE.json configs[2] (batch 8 per GPU).  Scenes are independent,
so ranks shard them with no data-path collective ("scaling": "weak"); value = scenes processed
by all ranks / max-over-ranks wall time.

Rank 0 prints ONE JSON line with the metric plus
  "roofline":     the dominant kernel of this run, timed with HIP events inside the timed region;
  "cpu_baseline": the same forward on the host CPU through the C oracle (oracle/, kind "port"),
                  on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

TRAIN_GC_INTERVAL = 50     # RefineTrainer(gc_interval=...): iterations between the trainer's own garbage collections
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (same as the fp32 vector peak)
SCORENET_GFLOP_PER_SCENE = {25600: 148.27, 51200: 180.20}  # SURVEY.md §8(d), 2*MAC of every 1x1 conv


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps (each timed run starts from an idle pipeline: its ~20 ms fill + drain is inside the timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="scenes per GPU per step (configs[2]: 8)")
    ap.add_argument("--points", type=int, default=25600)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="scenes per step over ALL ranks (configs[3]: 16 on 2 or 4 GPUs, configs[4]: 32 x 51 200 points on 8); "
                         "the per-rank batch is derived from it (overrides --batch / --train-batch)")
    ap.add_argument("--cpu-scenes", type=int, default=4, help="scenes timed for cpu_baseline (0 = skip)")
    ap.add_argument("--score-only", action="store_true", help="configs[1]: ScoreNet forward only")
    ap.add_argument("--latency-runs", type=int, default=5,
                    help="batch-1 forwards timed AFTER the timed region for latency_ms_single_scene (0 = skip, e.g. under "
                         "rocprofv3 so that the trace holds the bench's launches only)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket every n-th call of each native op (per shape) with HIP events; bracketing all calls costs ~2 %")
    ap.add_argument("--train", action="store_true",
                    help="configs[3]: the reference's training iteration (forward with labels, losses, backward, two Adam "
                         "steps, one flat gradient all-reduce per network over RCCL) instead of the forward pipeline")
    ap.add_argument("--exclusive-steps", type=int, default=48,
                    help="with --mlp-streams > 1: extra steps (outside the timed region) with one feature-stage stream for "
                         "per-kernel accounting (roofline_exclusive); 0 = skip")
    ap.add_argument("--gc-interval", type=int, default=TRAIN_GC_INTERVAL,
                    help="--train: iterations between the trainer's own garbage collections (RefineTrainer(gc_interval=...))")
    ap.add_argument("--mlp-streams", type=int, default=1, help="feature-stage streams (batches whose MFMA kernels may overlap): 2 is ~6 %% faster (875 scenes/s) but "
                         "time-shares the launches, so per-kernel durations stop being a kernel property (DESIGN.md par. 7)")
    ap.add_argument("--fps-streams", type=int, default=2, help="level-1 sampling launches in flight")
    ap.add_argument("--fps-group", type=int, default=0, help="batches whose level-1..3 sampling shares one launch (0: 64 scenes' worth, at most 8 batches)")
    ap.add_argument("--first-launch-groups", type=int, default=4,
                    help="sampling groups the FIRST sampling launch of a run takes (it finds the chip idle); the line reports "
                         "the resulting look-ahead (config.sampling_lookahead_batches) and value_no_lookahead beside the headline")
    ap.add_argument("--train-steps", type=int, default=24,
                    help="forward bench only: training iterations (configs[3] shapes, batch --train-batch) timed AFTER the timed "
                         "region for the line's \"train\" object (5 warm-up iterations first); 0 = skip")
    ap.add_argument("--single-rank-group", action="store_true",
                    help="--gpus 1 only: initialise a ONE-rank RCCL process group anyway and run what a multi-GPU job runs on it "
                         "(config.collective, the gradient bucket's all-reduce): scripts/scale_driver.sh at N = 1")
    ap.add_argument("--train-accounting-steps", type=int, default=4,
                    help="replayed training iterations carry no per-launch events: this many extra EAGER iterations after the "
                         "timed region give the train roofline its launch durations (0: none, e.g. under a profiler)")
    ap.add_argument("--train-batch", type=int, default=8)
    ap.add_argument("--train-graphs", choices=("auto", "on", "off"), default="auto",
                    help="replay the fixed-shape part of a training iteration as hipGraphs (train_step._TrunkGraphs); auto = the module switch")
    ap.add_argument("--train-distinct-batches", type=int, default=4,
                    help="training measurement: distinct batches the iterations cycle through (labels and scenes resident)")
    ap.add_argument("--train-timeline-steps", type=int, default=3,
                    help="training measurement: extra iterations under torch.profiler AFTER the timed ones for gpu_idle_ms_per_step (0 = skip)")
    ap.add_argument("--no-lookahead-steps", type=int, default=20,
                    help="steps of the extra pass AFTER the timed region with one sampling launch per batch and no enlarged "
                         "first launch (value_no_lookahead); 0 = skip")
    ap.add_argument("--real-density-steps", type=int, default=20,
                    help="steps of the extra pass AFTER the timed region over density-matched scenes (synthetic.make_scene(density="
                         "'real'): level-1 neighbourhood sizes like the reference's own clouds) for value_real_density; 0 = skip")
    ap.add_argument("--split-products-steps", type=int, default=20,
                    help="steps timed AFTER the timed region with the fenced split-products experiment on (fused.SPLIT_PRODUCTS: the level-1 "
                         "block on the bf16 matrix pipe, fp32-faithful) for value_split_products; 0 = skip.  Never part of `value`")
    ap.add_argument("--lookahead", type=int, default=3, help="batches whose region stage may be pending (pipeline depth)")
    ap.add_argument("--geometry-ahead", type=int, default=1, help="batches whose ball-query / 3-NN geometry is queued ahead of the feature stage")
    ap.add_argument("--graphs", choices=("auto", "on", "off"), default="auto",
                    help="replay the geometry + feature stages of a batch as hipGraphs: auto = batches of at most 4 x 25 600 "
                         "points (launch-bound shapes; the default workload, 8 x 25 600, is not one of them)")
    ap.add_argument("--distinct-batches", type=int, default=8,
                    help="distinct batches of scenes the steps cycle through (SURVEY 8d: scene i of a run uses seed 1000+i; "
                         "step k of the timed region feeds batch k %% this; 1 = the same 8 scenes every step, as rounds 1-3)")
    ap.add_argument("--no-region-calibration", action="store_true",
                    help="keep the region head's purely seeded weights (rounds 1-3): fewer than two closing boxes hold points, "
                         "so the reference's own `if len(gripper_mask) >= 2` skips the refine network")
    ap.add_argument("--set", action="append", default=[], metavar="module.NAME=0|1",
                    help="A/B measurement only: flip a module-level switch of the package before the run, e.g. "
                         "--set fused.FP_HEAD_INTERP=0 (reported in config.switches)")
    return ap.parse_args()


class OpTimer:
    """Brackets every native-op call of the product package with HIP events on the stream the
    kernel is launched on (torch's current stream), without synchronising."""

    def __init__(self, every=1):
        self.records = []  # (name, meta, start_event, end_event)
        self.enabled = False
        self.every = max(1, int(every))   # bracket every n-th call of an op (event records are not free, see --time-every)
        self.calls = {}
        self.critical_streams = None      # stream handles of the stage that bounds the step (ForwardPipeline's feature stage)
        self.critical = set()             # (name, shape) keys launched on one of them

    def wrap(self, module, name, meta_fn):
        orig = getattr(module, name)

        def timed(*a, **k):
            if not self.enabled or torch.cuda.is_current_stream_capturing():   # (a hipGraph capture: no timing events inside)
                return orig(*a, **k)
            key = (name, meta_fn(*a, **k))
            n = self.calls.get(key, 0)
            self.calls[key] = n + 1
            if n == 0 and self.critical_streams is not None and torch.cuda.current_stream().cuda_stream in self.critical_streams:
                self.critical.add(key)
            if n % self.every:
                return orig(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()  # records on torch's CURRENT stream == the stream the kernel is launched on
            out = orig(*a, **k)
            e.record()
            self.records.append((key[0], key[1], s, e))
            return out
        setattr(module, name, timed)

    def summary(self):
        """(name, shape) -> (total ms, calls) over ALL calls of the timed region: the mean of the bracketed calls (every
        ``self.every``-th one of each shape) times the number of calls."""
        agg = {}
        for name, meta, s, e in self.records:
            key = (name, meta)
            ms = s.elapsed_time(e)
            tot, cnt = agg.get(key, (0.0, 0))
            agg[key] = (tot + ms, cnt + 1)
        return {key: (tot / cnt * self.calls[key], self.calls[key]) for key, (tot, cnt) in agg.items()}


def install_timers(timer):
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    from regnet_for_3d_grasping_amd import region_ops
    import regnet_for_3d_grasping_amd.get_regiondataset as grd
    import regnet_for_3d_grasping_amd.gripper_region_network as grn
    ext = fn.pn2_ext
    timer.wrap(ext, "farthest_point_sample", lambda p, m, chain=None: ("B%d N%d M%d" % (p.size(0), p.size(2), m)))
    timer.wrap(ext, "ball_query", lambda p, c, r, k: ("B%d N%d M%d K%d" % (p.size(0), p.size(2), c.size(2), k)))
    timer.wrap(ext, "point_search", lambda q, k, n: ("B%d Q%d K%d" % (q.size(0), q.size(2), k.size(2))))
    timer.wrap(ext, "group_points_forward", lambda x, i: ("B%d C%d M%d K%d" % (x.size(0), x.size(1), i.size(1), i.size(2))))
    timer.wrap(ext, "interpolate_forward", lambda x, i, w: ("B%d C%d M%d N%d" % (x.size(0), x.size(1), x.size(2), i.size(1))))
    timer.wrap(region_ops, "radius_candidates", lambda pc, c, r: ("B%d N%d C%d" % (pc.size(0), pc.size(1), c.size(1))))
    timer.wrap(region_ops, "box_candidates", lambda g, *a: ("n%d G%d" % (g.size(0), g.size(1))))
    timer.wrap(region_ops, "gather_max", lambda f, r: ("R%d G%d F%d" % (r.size(0), r.size(1), f.size(1))))
    from regnet_for_3d_grasping_amd import fused
    for name, meta in getattr(fused, "TIMED_OPS", {}).items():
        timer.wrap(fused, name, meta)
    assert grd.region_ops is region_ops and grn.region_ops is region_ops


def algorithmic_work(name, meta):
    """(bound, units) of one launch: HBM bytes for the scan/gather kernels (SURVEY.md §8d, int64
    indices, op-API granularity), flops for the shared-MLP contraction."""
    d = {}
    for tok in meta.split():
        i = 0
        while i < len(tok) and not tok[i].isdigit():
            i += 1
        d[tok[:i]] = int(tok[i:])
    if name == "farthest_point_sample":
        return "hbm", d["B"] * (12 * d["N"] + 8 * d["M"])
    if name == "ball_query":
        return "hbm", d["B"] * (12 * d["N"] + 12 * d["M"] + 8 * d["M"] * d["K"] + 8 * d["M"])
    if name == "point_search":
        return "hbm", d["B"] * (12 * d["Q"] + 12 * d["K"] + 36 * d["Q"])
    if name == "group_points_forward":
        return "hbm", d["B"] * (8 * d["M"] * d["K"] + 8 * d["C"] * d["M"] * d["K"])
    if name == "interpolate_forward":
        return "hbm", d["B"] * (4 * d["C"] * d["M"] + 36 * d["N"] + 4 * d["C"] * d["N"])
    if name == "radius_candidates":
        return "hbm", d["B"] * (24 * d["N"] + 4 * d["C"] * d["N"])
    if name == "gather_max":
        return "hbm", d["R"] * d["G"] * d["F"] * 4 + d["R"] * d["F"] * 4
    if name == "box_candidates":
        return "hbm", d["n"] * d["G"] * 28
    if "flop" in d:
        return "mfma", d["flop"]
    return "hbm", 0


# host wrapper -> device kernel it launches (for grouping launches into kernel families)
def _mlp_layer_kernel(d):
    """Device kernel a fused.mlp_layer call launches, at rocprofv3's granularity (one name per template instantiation;
    the choice mirrors csrc/mlp.hip:launch_gemm2): skinny problems (<= 1024 rows: the region heads) and K < 17 take the
    round-1 kernel, everything else gemm2_kernel -- 256 x 128 tiles (pooling or not) or 128 x 128."""
    P, K, N = d.get("P", 0), d.get("K", 0), d.get("N", 0)
    if P <= 1024 or K < 17:
        return "mlp_gemm_kernel<0>"
    nt = (N + 127) // 128
    t256, t128, kpad = ((P + 255) // 256) * nt, ((P + 127) // 128) * nt, (K + 15) // 16 * 16
    if N > 128 and t256 >= 1024:
        tile = "256,128"
    elif t128 >= 1024 and (kpad < 256 or d.get("pool")):   # (two or more 128-wide slabs of K: the slab-accumulating tile)
        tile = "128,128"
    else:
        tile = "64,128"
    return "gemm2_kernel<%s%s>" % (tile, ",pool" if d.get("pool") else "")


KERNEL_OF = {"native_fwd": "tgemm_kernel", "native_dgrad": "tgemm_kernel", "native_wgrad": "tgemm_kernel",
             "native_fwd_bnrelu": "tgemm_kernel", "native_wgrad_bnrelu": "tgemm_kernel",   # (the family: tgemm_kernel / tgemm_stream_kernel)
             "mlp_layer": _mlp_layer_kernel, "sa_layer1": "mlp_gemm_kernel", "sa_layer12": "mlp_gemm_kernel",
             "sa_premul_layer": "mlp_gemm_kernel<3>", "sa_chain3": "sa_chain_kernel", "fp_head_chain": "fp_head_chain_kernel", "fp_head_chain_interp": "fp_head_chain_kernel", "sa_premul_chain": "sa_premul_chain_kernel", "sa3_premul_chain": "sa3_premul_chain_kernel",
             "farthest_point_sample": "fps_kernel", "ball_query": "ball_query_kernel",
             "point_search": "three_nn_kernel", "radius_candidates": "radius_group_kernel",
             "box_candidates": "box_crop_kernel", "gather_max": "gather_max_kernel"}


def pmc_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections applied; produced by
    scripts/collect_pmc.py and committed under profiles/).  None when no such file exists."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def _library_kernels():
    """Kernel names (rocprofv3's demangled spelling, template arguments kept, parameter list dropped) defined by the library
    that is loaded now; empty set when ``nm`` is not usable."""
    import subprocess
    from regnet_for_3d_grasping_amd import _lib
    try:
        syms = subprocess.run(["nm", "-C", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return set()
    have = set()
    for line in syms.splitlines():
        parts = line.split(" ", 2)
        if len(parts) == 3:
            name = parts[2].strip()
            name = name[5:] if name.startswith("void ") else name
            depth = 0
            for i, ch in enumerate(name):
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    name = name[:i]
                    break
            have.add(name)
    return have


def rocprof_launch(kernel, batch, points):
    """(average launch duration in ms, source file, note) of ``kernel``'s family in the committed rocprofv3 --kernel-trace
    --stats summary of this same command (profiles/rocprof_launch_ms.json, written by scripts/rocprof_summary.py --json from
    the trace whose text form is the file named in its "source").  The duration is handed out only when the trace is about
    THIS run: same scenes per batch and points per scene (the file's "workload"), and every kernel name it recorded for the
    family still defined by the loaded library; otherwise (None, source, why)."""
    path = os.path.join(REPO, "profiles", "rocprof_launch_ms.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        fam = doc["families"][kernel]
    except (OSError, ValueError, KeyError):
        return None, None, "no committed trace entry for this kernel family"
    src = doc.get("source")
    wl = doc.get("workload") or {"batch": 8, "points": 25600}      # (files written before the field existed: the default command)
    if (int(wl.get("batch", 0)), int(wl.get("points", 0))) != (int(batch), int(points)):
        return None, src, "stale: the committed trace is of batch %s x %s points, this run of %d x %d" % (
            wl.get("batch"), wl.get("points"), batch, points)
    have = _library_kernels()
    missing = [n for n in fam.get("kernel_names", []) if have and n not in have]
    if missing:
        return None, src, "stale: the loaded library no longer defines %s" % ", ".join(missing)
    return fam["avg_launch_ms"], src, None


def traffic_source():
    """Is profiles/pmc_traffic.json still about THIS library?  Every kernel name the PMC passes recorded (``kernel_names`` per
    family, scripts/collect_pmc.py) must be a kernel of the library that is loaded now (``nm -C`` of the .so: the kernel
    handles carry rocprofv3's demangled spelling).  -> {"file", "stale": True / False / None (no names recorded or no nm),
    "missing": [...]}."""
    path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    out = {"file": "profiles/pmc_traffic.json", "stale": None, "missing": []}
    try:
        with open(path) as f:
            recorded = sorted({n for v in json.load(f).values() if isinstance(v, dict) for n in v.get("kernel_names", [])})
    except (OSError, ValueError):
        return out
    have = _library_kernels()
    if not recorded or not have:
        return out
    out["missing"] = [n for n in recorded if n not in have]
    out["stale"] = bool(out["missing"])
    return out


def cpu_baseline(args, n_scenes, gpu_models=None):
    """The same forward on the host CPU: the host-side mirror driven by the C oracle (OpenMP) and
    torch-CPU 1x1 convs.  Bounded sample: ``n_scenes`` scenes of the bench workload, batch 1.
    With ``gpu_models`` (the bench's networks) the CPU side runs THEIR weights and the first scene is also pushed
    through the GPU path: the returned ``parity`` is the metric's "fp max-abs-err vs ref" on identical inputs."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    threads = torch.get_num_threads()
    score_net, region_net = pipeline.build_models("cpu")
    pc0 = synthetic.make_batch(1000, 1, args.points)
    first_out = None
    from oracle import pn2_ext_oracle
    with oracle_backend(), pn2_ext_oracle.native_build() as native:
        if gpu_models is not None:
            score_net.load_state_dict({k: v.cpu() for k, v in gpu_models[0].state_dict().items()})
            region_net.load_state_dict({k: v.cpu() for k, v in gpu_models[1].state_dict().items()})
        else:
            synthetic.calibrate_score_head(score_net, pc0)
        np.random.seed(0)
        t0 = time.perf_counter()
        for i in range(n_scenes):
            pc = pc0 if i == 0 else synthetic.make_batch(1000 + i, 1, args.points)
            t1 = time.perf_counter()
            out = pipeline.forward_scenes(score_net, region_net, pc, with_region=not args.score_only)
            if i == 0:
                first = time.perf_counter() - t1
                first_out = out
        dt = time.perf_counter() - t0
    parity = None
    if gpu_models is not None:
        dev = next(gpu_models[0].parameters()).device
        np.random.seed(0)   # the region stage draws from numpy's global stream: same seed as the CPU side's first scene
        got = pipeline.forward_scenes(gpu_models[0], gpu_models[1], pc0.to(dev), with_region=not args.score_only)
        torch.cuda.synchronize()
        ref_f, got_f = first_out["all_feature"], got["all_feature"].cpu()
        parity = {"score_max_abs_err": float((got["score"].cpu() - first_out["score"]).abs().max()),
                  "feature_max_rel_err": float(((got_f - ref_f).abs() / (1.0 + ref_f.abs())).max()),
                  "tolerance": 1e-4, "scene": "seed 1000, %d pts, batch 1, same weights on both sides" % args.points}
        if "center_pc_index" in got:
            same = (torch.equal(got["center_pc_index"].cpu(), first_out["center_pc_index"])
                    and torch.equal(got["pc_group_index"].cpu(), first_out["pc_group_index"])
                    and got["next_grasp"].shape == first_out["next_grasp"].shape)
            # centre selection thresholds the scores at 0.5: a score within fp32 noise of it may flip a centre
            parity["region_indices_equal"] = bool(same)
            parity["grasp_max_abs_err"] = (float((got["next_grasp"].cpu() - first_out["next_grasp"]).abs().max())
                                           if same else None)
    return {"value": n_scenes / dt, "unit": "scenes/s", "cores": threads, "kind": "port",
            "sample": "%d scene(s) x %d pts, batch 1, eval forward (%s), oracle C kernels (OpenMP, %s) + torch-CPU convs, "
                      "%.1f s total; host %s" % (n_scenes, args.points,
                                                 "ScoreNet" if args.score_only else "ScoreNet+grouping+GRN+refine",
                                                 "-O3 -march=native, built on this host" if native else "-O2 portable build", dt,
                                                 _cpu_model()),
            "parity": parity}


class HostBlockedTime:
    """Context manager: seconds the calling thread spends BLOCKED waiting for the device inside the package's synchronising
    reads (Tensor.cpu / .item / .tolist of GPU tensors, Event.synchronize, Stream.synchronize, torch.cuda.synchronize)."""

    def __init__(self):
        self.seconds, self.calls, self._saved = 0.0, 0, []
        self.log = []       # (name, seconds) of every blocking call, in order (measure_train slices it per iteration)

    def _wrap(self, owner, name, needs_cuda_self):
        orig = getattr(owner, name)
        outer = self

        def timed(*a, **k):
            if needs_cuda_self and not (a and getattr(a[0], "is_cuda", False)):
                return orig(*a, **k)
            t = time.perf_counter()
            try:
                return orig(*a, **k)
            finally:
                d = time.perf_counter() - t
                outer.seconds += d
                outer.calls += 1
                outer.log.append((name, d))
        self._saved.append((owner, name, orig))
        setattr(owner, name, timed)

    def __enter__(self):
        for name in ("cpu", "item", "tolist"):
            self._wrap(torch.Tensor, name, True)
        self._wrap(torch.cuda.Event, "synchronize", False)
        self._wrap(torch.cuda.Stream, "synchronize", False)
        self._wrap(torch.cuda, "synchronize", False)
        return self

    def __exit__(self, *exc):
        for owner, name, orig in reversed(self._saved):
            setattr(owner, name, orig)
        self._saved = []


def gpu_timeline(step_fn, n):
    """Run ``step_fn(k)`` for k in range(n) under torch.profiler (device activities only) and reduce the kernels' start / end
    stamps of ALL streams to the union of busy intervals: -> {steps, span_ms_per_step, busy_ms_per_step, idle_ms_per_step,
    kernels_per_step, gaps_over_20us_per_step}, or {"error": ...} when the profiler is not usable on this box."""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            t0 = time.perf_counter()
            for k in range(n):
                step_fn(k)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        spans = []
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower():
                tr = ev.time_range
                if tr.end > tr.start:
                    spans.append((tr.start, tr.end))
        if not spans:
            return {"error": "no device activity recorded"}
        spans.sort()
        busy, gaps, (cur_s, cur_e) = 0.0, 0, spans[0]
        for s0, e0 in spans[1:]:
            if s0 > cur_e:
                busy += cur_e - cur_s
                gaps += (s0 - cur_e) > 20.0
                cur_s, cur_e = s0, e0
            else:
                cur_e = max(cur_e, e0)
        busy += cur_e - cur_s
        span = max(e for _, e in spans) - spans[0][0]
        return {"steps": n, "wall_ms_per_step_under_profiler": round(wall / n * 1e3, 3),
                "span_ms_per_step": round(span / n / 1e3, 3), "busy_ms_per_step": round(busy / n / 1e3, 3),
                "idle_ms_per_step": round((span - busy) / n / 1e3, 3), "kernels_per_step": round(len(spans) / n, 1),
                "gaps_over_20us_per_step": round(gaps / n, 1)}
    except Exception as exc:      # a side measurement: never takes the line down
        return {"error": repr(exc)[:200]}


_REAL_STDOUT = None


def guard_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL prints a five-line version banner to the
    C-level stdout of rank 0 when its first communicator comes up, and C stdio flushes it at exit, AFTER the line.  So file
    descriptor 1 is pointed at stderr for the whole run and the line is written to the real stdout by ``emit`` at the end."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    """The run's one JSON line, to the process's real stdout."""
    data = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def train_phases(marks):
    """RefineTrainer.phase_marks of the timed iterations (after a synchronisation) -> per-iteration durations and, for
    replayed iterations, the median split of an iteration of the trunk's stream: forward / head backward / wait for the
    region stage and its backward / trunk backward / all-reduce + optimizers; ``busy_ms`` = the four pieces of own work."""
    if not marks:
        return {}
    iters = []
    for a, b in zip(marks[:-1], marks[1:]):
        iters.append(a["start"].elapsed_time(b["start"]))
    iters.append(marks[-1]["start"].elapsed_time(marks[-1]["end"]))
    out = {"iteration_ms": [round(x, 3) for x in iters], "iteration_median_ms": round(float(np.median(iters)), 3)}
    full = [m for m in marks if all(k in m for k in ("forward", "head", "joined", "trunk", "end"))]
    if full:
        def med(a, b):
            return round(float(np.median([m[a].elapsed_time(m[b]) for m in full])), 3)
        ph = {"forward_ms": med("start", "forward"), "head_backward_ms": med("forward", "head"),
              "region_wait_ms": med("head", "joined"), "trunk_backward_ms": med("joined", "trunk"),
              "allreduce_optimizer_ms": med("trunk", "end"), "iterations": len(full)}
        ph["busy_ms"] = round(ph["forward_ms"] + ph["head_backward_ms"] + ph["trunk_backward_ms"] + ph["allreduce_optimizer_ms"], 3)
        out["phases"] = ph
    return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return "%s x%d" % (line.split(":", 1)[1].strip(), os.cpu_count())
    except OSError:
        pass
    return "unknown x%d" % (os.cpu_count() or 0)


def _pts(n):
    """25600 -> '25 600' (BASELINE.json writes the metric's point count that way)."""
    return "{:,}".format(int(n)).replace(",", " ")


def measure_train(args, rank, world, dev, steps, warmup, batch):
    """configs[3-4]: ``steps`` training iterations of ``train_step.RefineTrainer`` on this rank's scenes (synthetic labels,
    fresh random-init networks).  Returns the fields of a bench line (rank 0) or None."""
    from regnet_for_3d_grasping_amd import pipeline, sharding, synthetic
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    from regnet_for_3d_grasping_amd.train_step import RefineTrainer
    B, N = batch, args.points
    # the iterations cycle through `distinct` batches of this rank's shard (scene i of a run uses seed 1000 + i;
    # iteration k feeds global batch k % distinct), all resident in HBM, labels parsed, before the timed region
    distinct = max(1, args.train_distinct_batches)
    batches = []
    for k in range(distinct):
        seeds = sharding.scene_seeds(rank, world, B, step=k)
        pc_k = torch.from_numpy(np.stack([synthetic.make_scene(s_, N) for s_ in seeds], 0))
        records_k = [synthetic.make_grasp_labels(pc_k[b].numpy(), 50 + s_) for b, s_ in enumerate(seeds)]
        target_k = torch.from_numpy(np.random.default_rng([2, seeds[0]]).uniform(0, 1, (B, N)).astype(np.float32)).to(dev)
        batches.append((pc_k.to(dev), target_k, records_k))
    score_net = ScoreNetwork(training=True)
    score_net.load_state_dict(synthetic.seeded_state_dict(score_net, 7))
    region_net = GripperRegionNetwork(training=True, group_num=pipeline.GROUP_NUM, gripper_num=pipeline.GRIPPER_NUM,
                                      grasp_score_threshold=pipeline.GRASP_SCORE_THRESHOLD, radius=pipeline.DEPTH,
                                      reg_channel=pipeline.REG_CHANNEL)
    region_net.load_state_dict(synthetic.seeded_state_dict(region_net, 11))
    synthetic.set_region_head_affine(region_net)   # decoded grasps hold points: the refine losses run on real rows
    import gc
    gc_was_on = gc.isenabled()
    trainer = RefineTrainer(score_net.to(dev), region_net.to(dev), pipeline.PARAMS, pipeline.GRIPPER_PARAMS,
                            gc_interval=args.gc_interval, graphs={"auto": None, "on": True, "off": False}[args.train_graphs])
    np.random.seed(rank)
    # HIP events around the native 1x1-convolution kernels (forward / input gradient / weight gradient: the MFMA work of
    # the iteration) on the stream they are launched on, inside the timed region
    from regnet_for_3d_grasping_amd import conv1x1_train
    timer = OpTimer(every=args.time_every)
    originals = {name: getattr(conv1x1_train, name) for name in conv1x1_train.TIMED_OPS}
    for name, meta in conv1x1_train.TIMED_OPS.items():
        timer.wrap(conv1x1_train, name, meta)

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # steady state of a training loop: the geometry (FPS / ball query / 3-NN: xyz only) of the NEXT batch is enqueued
    # on a side stream before each iteration, as a data loader with one batch of look-ahead would
    it = 0
    ahead = trainer.prefetch(batches[0][0])
    for _ in range(warmup):
        nxt = trainer.prefetch(batches[(it + 1) % distinct][0])
        pc, target, records = batches[it % distinct]
        trainer.step(pc, target, records, plan=ahead)
        ahead, it = nxt, it + 1
    fence()
    timer.enabled = True
    blocked = HostBlockedTime()
    region_steps = refine_steps = 0
    allreduce_ms = []
    trainer.phase_marks = []         # HIP events on the trunk's stream at the phase boundaries of every iteration
    host_marks, blocked_at = [], []
    replays_before = trainer.graph_replays
    with blocked:
        t0 = time.perf_counter()
        for _ in range(steps):
            host_marks.append(time.perf_counter())
            blocked_at.append(len(blocked.log))
            nxt = trainer.prefetch(batches[(it + 1) % distinct][0])
            pc, target, records = batches[it % distinct]
            loss, parts = trainer.step(pc, target, records, plan=ahead)
            ahead, it = nxt, it + 1
            region_steps += "region_error" not in parts
            refine_steps += parts.get("refine") is not None
            if trainer.bucket is not None and trainer.bucket.last_allreduce_ms() is not None:
                allreduce_ms.append(trainer.bucket.last_ms)      # (the most recent gradient all-reduce whose events completed)
        host_marks.append(time.perf_counter())
        blocked_at.append(len(blocked.log))
        t_enqueued = time.perf_counter() - t0
        fence()
        dt_local = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt_local, dev)
    if trainer.bucket is not None and trainer.bucket.last_allreduce_ms() is not None:
        allreduce_ms.append(trainer.bucket.last_ms)          # the last iteration's (its events have completed now)
    timer.enabled = False
    marks, trainer.phase_marks = trainer.phase_marks, None
    replays = trainer.graph_replays - replays_before
    accounting_steps = 0
    if replays and args.train_accounting_steps > 0:
        # the timed iterations replayed hipGraphs: no wrapper ran.  The contractions' launch durations (the `roofline` object)
        # come from a few extra EAGER iterations outside the timed region -- the same launches issued one by one
        accounting_steps, saved = args.train_accounting_steps, trainer.graphs
        trainer.graphs = False
        for k in range(accounting_steps + 2):
            # (two eager iterations first, unbracketed: the allocator's cache of the eager path -- 30 GB of activations -- may
            # have been trimmed since the warm-up, and an iteration that waits for hipMalloc is paced by the host: events
            # around a launch then time the launching thread, not the kernel)
            if k == 2:
                torch.cuda.synchronize()
                timer.enabled = True
            nxt = trainer.prefetch(batches[(it + 1) % distinct][0])
            trainer.step(*batches[it % distinct], plan=ahead)
            ahead, it = nxt, it + 1
        torch.cuda.synchronize()
        timer.enabled = False
        trainer.graphs = saved
    for name, fn in originals.items():
        setattr(conv1x1_train, name, fn)
    # per-iteration figures, all from this (un-profiled) timed region: the trunk stream's cadence and phases from the HIP
    # events, the launching thread's time per iteration, and what each blocking read of an iteration waited
    phases = train_phases(marks)
    host_iter = [round((b - a) * 1e3, 3) for a, b in zip(host_marks[:-1], host_marks[1:])]
    reads = [blocked.log[a:b] for a, b in zip(blocked_at[:-1], blocked_at[1:])]
    n_reads = max((len(r) for r in reads), default=0)
    read_wait = []
    for j in range(n_reads):
        col = [r[j] for r in reads if len(r) > j]
        read_wait.append({"call": col[0][0], "median_ms": round(float(np.median([d for _, d in col])) * 1e3, 3),
                          "max_ms": round(max(d for _, d in col) * 1e3, 3)})
    # where the GPU has nothing to run: a few MORE iterations under torch's profiler (kernel start / end stamps of every
    # stream -> union of busy intervals), outside the timed region, with the same one-batch geometry look-ahead
    state = {"ahead": ahead, "it": it}

    def profiled_step(_k):
        nxt_ = trainer.prefetch(batches[(state["it"] + 1) % distinct][0])
        trainer.step(*batches[state["it"] % distinct], plan=state["ahead"])
        state["ahead"], state["it"] = nxt_, state["it"] + 1

    gpu_idle = gpu_timeline(profiled_step, args.train_timeline_steps) if args.train_timeline_steps > 0 else None
    # what the trainer's own collection (every TRAIN_GC_INTERVAL iterations) costs: one collection timed here, amortised below
    t1 = time.perf_counter()
    gc.collect()
    gc_ms = (time.perf_counter() - t1) * 1e3
    if gc_was_on:
        gc.enable()
    if rank != 0:
        return None
    _, roofline = roofline_of(timer.summary(), accounting_steps or steps, B)
    if roofline and accounting_steps:
        roofline["accounting_note"] = ("the %d timed iterations replayed hipGraphs; launch durations are from %d extra eager "
                                       "iterations outside the timed region" % (steps, accounting_steps))
    grads = sum(p.numel() for net in (score_net, region_net) for p in net.parameters())
    return {"metric": "train scenes/sec (%s-pt ScoreNet+GRN+Refine training iteration)" % _pts(N),
            "value": round(B * steps * world / dt, 3), "unit": "scenes/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "roofline": roofline,
            # every timed iteration by itself (HIP events on the trunk's stream, iteration i = start mark i -> start mark i + 1,
            # the last one -> its own end mark): a stall shows as an outlier here, a slow host as a level shift of
            # host_iteration_ms; the phases split the trunk stream's time into its own work and the wait for the region stage
            "ms_per_step_median": phases.get("iteration_median_ms"), "iteration_ms": phases.get("iteration_ms"),
            "host_iteration_ms": host_iter, "trunk_stream": phases.get("phases"),
            "graph_replays": replays,
            "readback_wait_ms": read_wait,
            # the launching thread: wall time of the timed iterations minus the time it was BLOCKED in a device->host read or
            # a synchronisation (waiting for the GPU); host_ms_per_step ~ ms_per_step means the iteration is paced by the host
            "host_ms_per_step": round((dt_local - blocked.seconds) / steps * 1e3, 3),
            "host_blocked_ms_per_step": round(blocked.seconds / steps * 1e3, 3),
            "host_blocking_reads_per_step": round(blocked.calls / steps, 1),
            "host_enqueue_done_ms_per_step": round(t_enqueued / steps * 1e3, 3),
            # union of all kernels' busy intervals over `gpu_timeline.steps` extra iterations (torch.profiler, after the timed region)
            "gpu_idle_ms_per_step": None if not gpu_idle else gpu_idle.get("idle_ms_per_step"),
            "gpu_timeline": gpu_idle,
            # event-timed duration of the iteration's single flat gradient all-reduce (RCCL, side stream); null at 1 GPU
            "allreduce_ms": round(sum(allreduce_ms) / len(allreduce_ms), 4) if allreduce_ms else None,
            # CPython's automatic cyclic collector is off during training (RefineTrainer(gc_interval=N) collects every N
            # iterations itself: left on, it stalls every ~10th iteration by 60-100 ms); the amortised cost is in this figure
            "gc": {"interval_iterations": args.gc_interval, "one_collection_ms": round(gc_ms, 2),
                   "ms_per_step_incl_amortised_gc": round(dt / steps * 1e3 + (gc_ms / args.gc_interval if steps < args.gc_interval else 0.0), 3)},
            "config": {"workload": "%s: training iteration (forward with labels, stage-2 + refine losses, backward, "
                                   "two Adam steps), %d-pt synthetic scenes, batch=%d per GPU" % (
                                       "configs[4]" if N == 51200 else "configs[3]", N, B),
                       "points": N, "batch_per_gpu": B, "global_batch": B * world, "distinct_batches": distinct,
                       "bucket_bytes": None if trainer.bucket is None else int(trainer.bucket.flat.numel()) * 4,
                       "parallelism": "dp%d: ONE flat fp32 gradient all-reduce per iteration (%d elements, both networks) over RCCL" % (world, grads),
                       "steps_with_region_losses": region_steps, "steps_with_refine_losses": refine_steps,
                       "last_loss": float(loss)}}


def run_train(args, rank, world, dev, collective=None):
    res = measure_train(args, rank, world, dev, args.steps, args.warmup, args.batch)
    if rank == 0:
        if collective is not None:
            res["config"]["collective"] = dict(collective, bucket_bytes=res["config"].pop("bucket_bytes", None),
                                               allreduce_ms=res["allreduce_ms"])
        emit(res)


def roofline_of(agg, steps, batch, critical=None):
    """OpTimer summary -> (kernel families, roofline object of the dominant one).  ``critical``: the (name, shape) keys that
    were launched on the stream of the stage that bounds the step (the pipeline's feature stage, whose queue is busy 98 % of a
    step: DESIGN.md par. 7); only their families can be the dominant kernel -- the sampling / grouping / region kernels of the
    other batches run beside it on their own streams and their event time is mostly waiting for a CU."""
    # kernel families: every mlp_layer / sa_layer1 launch is the same device kernel (mlp_gemm_kernel)
    fam = {}
    for (name, meta), (tot, cnt) in agg.items():
        bound, units = algorithmic_work(name, meta)
        key = KERNEL_OF.get(name, name)
        if callable(key):
            d = {}
            for tok in meta.split():
                i = 0
                while i < len(tok) and not tok[i].isdigit():
                    i += 1
                d[tok[:i]] = int(tok[i:])
            key = key(d)
        f = fam.setdefault(key, {"ms": 0.0, "launches": 0, "units": 0, "bound": bound, "critical": critical is None})
        f["critical"] = f["critical"] or (critical is not None and (name, meta) in critical)
        f["ms"] += tot
        f["launches"] += cnt
        f["units"] += units * cnt
    if not fam:
        return fam, None

    # kernels are told apart as rocprofv3 does (one name per template instantiation), so that the committed
    # --kernel-trace summary can be held against this line kernel by kernel
    # dominant = most GPU resource-time: a furthest-point-sampling launch keeps ONE CU per
    # scene busy (a latency chain running beside the MLPs), every other kernel fills the chip
    def cu_ms(k):
        if not fam[k]["critical"]:
            return 0.0
        if k == "fps_kernel":
            return fam[k]["ms"] * min(1.0, batch / 256.0)
        if k == "mlp_gemm_kernel<0>":   # the skinny split-K launches of the region heads: at most 256 of the chip's 1024
            return fam[k]["ms"] * 0.25  # workgroup slots, on the region stream (their event time is mostly queueing)
        return fam[k]["ms"]
    dom = max(fam, key=cu_ms)
    f = fam[dom]
    avg_s = f["ms"] / f["launches"] / 1e3
    per_launch = f["units"] / f["launches"]
    if f["bound"] == "hbm":
        achieved, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
    else:
        achieved, peak, unit = per_launch / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
    roofline = {"bound": f["bound"], "achieved": round(achieved, 4), "peak": peak, "unit": unit,
                "frac": round(achieved / peak, 6), "traffic": pmc_traffic(dom), "kernel": dom,
                "avg_launch_ms": round(f["ms"] / f["launches"], 4), "launches": f["launches"],
                "algorithmic_units_per_launch": round(per_launch),
                "families_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in
                                         sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:6]}}
    return fam, roofline


def sa_chain_executed_share(score_net, pcs):
    """Share of sa_chain_kernel's algorithmic flops (all 64 slots of every level-1 neighbourhood) that the kernel executes
    on these batches: neighbourhoods with <= 32 members run one 32-row point tile instead of two; two neighbourhoods with
    33..48 members that the kernel pairs (slots 8 g + w and 8 g + w + 4 of the processing order, csrc/sa_chain.hip) run three
    tiles instead of four.  Layers 2-3 are 49152 / 49920 of the block's flops.  None when the plan carries no counts."""
    from regnet_for_3d_grasping_amd import fused
    with torch.no_grad():
        counts = [score_net.plan(b)["sa"][0].get("count") for b in pcs]
    if not counts or any(c is None for c in counts):
        return None
    shares = [float((c <= 32).float().mean()) for c in counts]     # per distinct batch
    small = sum(shares) / len(shares)
    pshares = []
    for c in counts:
        cls = ((c.reshape(-1) > 32).to(torch.int8) + (c.reshape(-1) > 48).to(torch.int8))[fused.chain3_order(c)]
        full = cls[:cls.numel() // 8 * 8].view(-1, 2, 4)          # [workgroup][w < 4 | w >= 4][w & 3]
        pshares.append(float(2 * ((full[:, 0] == 1) & (full[:, 1] == 1)).sum()) / cls.numel())
    paired = sum(pshares) / len(pshares)
    mean_count = sum(float(c.float().mean()) for c in counts) / len(counts)
    return {"small_ball_share": {"mean": round(small, 4), "min": round(min(shares), 4), "max": round(max(shares), 4),
                                 "batches": len(shares)},
            "paired_ball_share": round(paired, 4), "mean_ball_count": round(mean_count, 2),
            "executed_share_of_algorithmic_flops": round(1.0 - (0.5 * small + 0.25 * paired) * 49152.0 / 49920.0, 4)}


def main():
    args = parse()
    guard_stdout()
    from regnet_for_3d_grasping_amd import sharding
    rank, local_rank, world = sharding.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # tests only (tests/test_gpu_bench_contract.py): all ranks on device 0 over gloo, to drive the multi-rank code paths of
    # this script on a one-GPU box (RCCL refuses two ranks on one device)
    one_device = os.environ.get("REGNET_BENCH_ONE_DEVICE_GLOO") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # one process per GPU on one host: keep each rank's CPU-side helpers (torch intra-op pool, OpenMP) to its share
        # of the cores, so that N ranks' region-stage host threads do not fight over them
        # (sharding.pin_rank: a contiguous share of the GPU's NUMA node, affinity mask + OpenMP / torch pools)
        pinned = sharding.pin_rank(local_rank, world, None if one_device else local_rank)
        # RCCL; forward: only the barrier + max-over-ranks of the contract.  The streams of the workload are bound first.
        if not args.train:
            from regnet_for_3d_grasping_amd import pipeline as _pl
            _pl.reserve_streams(dev, args.fps_streams, args.mlp_streams)
        sharding.init("gloo" if one_device else "nccl", dev, reserve=("train",) if args.train else ("forward", "train"))
        import torch.distributed as dist
        if not one_device and str(dist.get_backend()) != "nccl":
            raise SystemExit("bench.py --gpus %d: the process group's backend is %r, not nccl (= RCCL on ROCm)" % (world, dist.get_backend()))
    # what the process group really is (backend, world size, RCCL version, a startup all-reduce proving `world` distinct
    # ranks / devices, the gradient-bucket-sized all-reduce's duration and bus bandwidth): config.collective of both lines
    if world == 1 and args.single_rank_group:
        # the multi-GPU code path on one GPU: a one-rank RCCL communicator, every collective of the job issued on it
        sharding.SINGLE_RANK_GROUP = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        pinned = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
        if not args.train:
            from regnet_for_3d_grasping_amd import pipeline as _pl
            _pl.reserve_streams(dev, args.fps_streams, args.mlp_streams)
        sharding.init("nccl", dev, reserve=("train",) if args.train else ("forward", "train"))
    collective = sharding.describe_collective(dev) if sharding.group_active() else None
    if collective is not None:
        collective["pinned_cores"] = len(pinned)
        if collective["distinct_ranks_by_allreduce"] != world or (not one_device and collective["distinct_devices"] != world):
            raise SystemExit("bench.py --gpus %d: the group holds %d distinct ranks on %d distinct devices" % (
                world, collective["distinct_ranks_by_allreduce"], collective["distinct_devices"]))
    import importlib
    for item in args.set:
        target, value = item.split("=")
        mod, name = target.rsplit(".", 1)
        module = importlib.import_module("regnet_for_3d_grasping_amd." + mod)
        current = getattr(module, name)
        assert isinstance(current, (bool, int)), target          # module switches are booleans or small integers
        setattr(module, name, value not in ("0", "false", "False") if isinstance(current, bool) else int(value))
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not a multiple of the %d ranks" % (args.global_batch, world))
        args.batch = args.train_batch = args.global_batch // world
    if args.train:
        run_train(args, rank, world, dev, collective)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    from regnet_for_3d_grasping_amd import pipeline, synthetic
    timer = OpTimer(args.time_every)
    install_timers(timer)

    score_net, region_net = pipeline.build_models(dev)
    # each rank owns its shard of independent scenes: rank r holds scenes (k*W + r)*B .. +B-1 of global batch k; the steps
    # cycle through `distinct` such batches, all resident in HBM before the timed region
    distinct = max(1, args.distinct_batches)
    pcs = []
    for k in range(distinct):
        seeds = sharding.scene_seeds(rank, world, args.batch, step=k)
        pcs.append(torch.from_numpy(np.stack([synthetic.make_scene(s_, args.points) for s_ in seeds], 0)).to(dev))
    pc = pcs[0]
    synthetic.calibrate_score_head(score_net, pc)
    np.random.seed(1234 + rank)
    region_calibration = None
    if not args.score_only and not args.no_region_calibration:
        # the seeded region head decodes grasps whose closing boxes are empty, which makes configs[2]'s third network a
        # no-op; calibrate its last BatchNorms on batch 0 (synthetic.calibrate_region_head, as tests/golden/s9_*)
        try:
            synthetic.calibrate_region_head(region_net, lambda: pipeline.forward_scenes(score_net, region_net, pc))
            region_calibration = "synthetic.calibrate_region_head on batch 0"
        except RuntimeError as exc:      # sparse test clouds (e.g. 6 144 points): no closing box holds > 5 points
            region_calibration = "stage-2 head calibrated, refine head seeded (%s)" % exc
        np.random.seed(1234 + rank)

    # One "step" = one batch through the whole forward hot path.  Steps are issued through
    # ForwardPipeline, which overlaps the geometry of the next batch, the MLPs of the current one
    # and the region stage of the previous one on three HIP streams (all work of the K timed
    # steps happens inside the timed region; the pipeline drains before the closing fence).
    pipe = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only, fps_streams=args.fps_streams,
                                    mlp_streams=args.mlp_streams, fps_group=args.fps_group,
                                    first_launch_groups=args.first_launch_groups, geometry_ahead=args.geometry_ahead,
                                    graphs={"auto": "auto", "on": True, "off": False}[args.graphs])

    refine_stats = []

    def feed(n):
        return (pcs[k % distinct] for k in range(n))

    def run_steps(n, stats=None):
        last = None
        for last in pipe.run(feed(n), max_pending_regions=args.lookahead):
            if stats is not None and "next_grasp" in last:
                sel = last.get("select_grasp_class")
                stats.append((int(last["next_grasp"].shape[0]), last.get("valid_crops"),
                              0 if sel is None else int(sel.shape[0])))
        return last

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    fence()
    # (read the handles only now: the HIP runtime binds a stream to a hardware queue when its handle is first used, and a
    # feature stream that is bound first -- before the pipeline's own first use of its streams -- ends up 8 % slower)
    timer.critical_streams = {m.cuda_stream for m in pipe.s_mlps}
    timer.enabled = True
    pipe.graph_replays = 0
    pipe.host_late_feature_stages = pipe.geometry_pending_at_enqueue = 0
    # CPython's cyclic collector inside the timed region (reported as config.host_gc): a generation-2 pass over a process
    # that holds a few hundred thousand torch objects stalls the launching thread for milliseconds
    import gc
    gc_log = {"n": [0, 0, 0], "ms": 0.0, "t": 0.0}

    def gc_watch(phase, info):
        if phase == "start":
            gc_log["t"] = time.perf_counter()
        else:
            gc_log["n"][info["generation"]] += 1
            gc_log["ms"] += (time.perf_counter() - gc_log["t"]) * 1e3
    gc.callbacks.append(gc_watch)
    t0 = time.perf_counter()
    out = run_steps(args.steps, refine_stats)
    fence()
    dt = time.perf_counter() - t0
    gc.callbacks.remove(gc_watch)
    timer.enabled = False
    main_summary = timer.summary() if rank == 0 else None
    dt = sharding.max_over_ranks(dt, dev)
    first_launch_batches = pipe.first_launch_batches
    first_launch_split = getattr(pipe, "first_launch_split", None)
    graph_replays = pipe.graph_replays
    graph_accounting_steps = 0
    if graph_replays and rank == 0:
        # the timed steps replayed hipGraphs: no per-launch events exist for their kernels.  The per-kernel accounting of the
        # line comes from a short pass of the SAME launches issued one by one (graphs off), outside the timed region.
        timer.records, timer.calls = [], {}
        pipe_e = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only, fps_streams=args.fps_streams,
                                          mlp_streams=1, fps_group=args.fps_group, graphs=False)
        timer.critical_streams = {m.cuda_stream for m in pipe_e.s_mlps}
        for _ in pipe_e.run(feed(4), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        timer.enabled = True
        graph_accounting_steps = max(8, min(args.steps, 48))
        for _ in pipe_e.run(feed(graph_accounting_steps), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        timer.enabled = False
        main_summary = timer.summary()

    no_lookahead = None
    if world == 1 and args.no_lookahead_steps > 0:
        # the same steps with ONE sampling launch per batch and a first launch like every other (the pipeline then reads
        # 3-4 batches ahead instead of up to 32): what the grouped sampling + enlarged first launch are worth, same
        # process, same scenes, outside the timed region
        pipe0 = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only,
                                         fps_streams=args.fps_streams, mlp_streams=args.mlp_streams, fps_group=1,
                                         first_launch_groups=1)
        for _ in pipe0.run(feed(max(2, args.warmup)), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in pipe0.run(feed(args.no_lookahead_steps), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        no_lookahead = args.batch * args.no_lookahead_steps / (time.perf_counter() - t1)

    split_products = None
    if world == 1 and args.split_products_steps > 0 and not args.score_only and args.points <= 102400:
        # The fenced ceiling experiment (DESIGN.md par. 10): the SAME pipeline, scenes and weights with sa_chain_kernel replaced by
        # csrc/sa_split.hip -- every operand as three bf16 pieces, six products on the bf16 matrix pipe, fp32 accumulation.  After
        # the timed region; its own value, its own parity figures (the bench scene's scores against the exact path's), never `value`.
        from regnet_for_3d_grasping_amd import fused as _fused
        with torch.no_grad():
            _, score_exact, _ = score_net(pcs[0])
        _fused.SPLIT_PRODUCTS = True
        try:
            with torch.no_grad():
                _, score_split, _ = score_net(pcs[0])
            pipe_s = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only, fps_streams=args.fps_streams,
                                              mlp_streams=args.mlp_streams, fps_group=args.fps_group,
                                              first_launch_groups=args.first_launch_groups, geometry_ahead=args.geometry_ahead,
                                              graphs={"auto": "auto", "on": True, "off": False}[args.graphs])
            was_enabled, timer.enabled = timer.enabled, False
            for _ in pipe_s.run((pcs[k % distinct] for k in range(max(2, args.warmup))), max_pending_regions=args.lookahead):
                pass
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in pipe_s.run((pcs[k % distinct] for k in range(args.split_products_steps)), max_pending_regions=args.lookahead):
                pass
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t1
            timer.enabled = was_enabled
        finally:
            _fused.SPLIT_PRODUCTS = False
        split_products = {"value": round(args.batch * args.split_products_steps / dt_s, 3), "unit": "scenes/s",
                          "ms_per_step": round(dt_s / args.split_products_steps * 1e3, 3), "steps": args.split_products_steps,
                          "what": "fused.SPLIT_PRODUCTS: the level-1 set-abstraction block (32.7 of a scene's 148.3 GFLOP) by "
                                  "sa_chain_split_kernel (v_mfma_f32_32x32x16_bf16 on three bf16 pieces per operand, six products, fp32 "
                                  "accumulation); every other kernel the exact-fp32 one",
                          "score_max_abs_diff_vs_exact_path": float((score_split - score_exact).abs().max()),
                          "positives_changed": int(((score_split > 0.5) != (score_exact > 0.5)).sum()),
                          # instruction ceilings measured on this chip (scripts/ablate/split_products.cpp): bf16 1 864 TFLOP/s / 6
                          "ceiling_tflops_fp32_equivalent": 310.6}

    real_density = None
    if world == 1 and args.real_density_steps > 0:
        # SURVEY 8(d) "real-density variant": the SAME pipeline configuration and weights over scenes whose level-1
        # neighbourhood sizes match the reference's own clouds (mean 49.5 members, 23 % <= 32, 49 % full: the uniform scenes
        # above have 30 % <= 32 and 19 % full) -- sa_chain_kernel skips less on them.  After the timed region.
        pcs_r = []
        for k in range(distinct):
            seeds = sharding.scene_seeds(rank, world, args.batch, step=k)
            pcs_r.append(torch.from_numpy(np.stack([synthetic.make_scene(s_, args.points, density="real") for s_ in seeds], 0)).to(dev))
        pipe_r = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only, fps_streams=args.fps_streams,
                                          mlp_streams=args.mlp_streams, fps_group=args.fps_group,
                                          first_launch_groups=args.first_launch_groups, geometry_ahead=args.geometry_ahead,
                                          graphs={"auto": "auto", "on": True, "off": False}[args.graphs])
        was_enabled, timer.enabled = timer.enabled, False
        for _ in pipe_r.run((pcs_r[k % distinct] for k in range(max(2, args.warmup))), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        stats_r = []
        t1 = time.perf_counter()
        for last in pipe_r.run((pcs_r[k % distinct] for k in range(args.real_density_steps)), max_pending_regions=args.lookahead):
            if "next_grasp" in last:
                stats_r.append(last.get("valid_crops"))
        torch.cuda.synchronize()
        dt_r = time.perf_counter() - t1
        timer.enabled = was_enabled
        real_density = {"value": round(args.batch * args.real_density_steps / dt_r, 3), "unit": "scenes/s",
                        "ms_per_step": round(dt_r / args.real_density_steps * 1e3, 3), "steps": args.real_density_steps,
                        "valid_crops_per_step": (None if not stats_r or any(v is None for v in stats_r)
                                                 else round(sum(stats_r) / len(stats_r), 1)),
                        "scenes": "synthetic.make_scene(seed, %d, density='real'), seeds as the headline's; histogram target "
                                  "tests/golden/real_density_hist.json (scripts/real_density_hist.py)" % args.points,
                        "level1_balls": sa_chain_executed_share(score_net, pcs_r)}
        del pcs_r, pipe_r

    exclusive = None
    if args.mlp_streams > 1 and world == 1 and args.exclusive_steps > 0:
        # per-kernel accounting without feature stages overlapping each other: a short extra pass (outside the timed
        # region) through a pipeline with ONE feature-stage stream, same scenes, same event bracketing
        timer.records, timer.calls = [], {}
        pipe1 = pipeline.ForwardPipeline(score_net, region_net, with_region=not args.score_only,
                                         fps_streams=args.fps_streams, mlp_streams=1)
        timer.critical_streams |= {m.cuda_stream for m in pipe1.s_mlps}
        for _ in pipe1.run(feed(4), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        timer.enabled = True
        t1 = time.perf_counter()
        for _ in pipe1.run(feed(args.exclusive_steps), max_pending_regions=args.lookahead):
            pass
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        timer.enabled = False
        exclusive = (timer.summary(), dt1)

    latency_ms = None
    if rank == 0 and world == 1 and args.latency_runs > 0:
        # latency of ONE scene through the whole forward (batch 1, nothing else in flight), outside the timed region:
        # the furthest-point-sampling chains (level 1: ~880 dependent rounds of ~6 exact picks on one CU) are ~5 of its ~8 ms
        one = pc[:1].contiguous()
        state = np.random.get_state()
        for _ in range(2):
            pipeline.forward_scenes(score_net, region_net, one, with_region=not args.score_only)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.latency_runs):
            pipeline.forward_scenes(score_net, region_net, one, with_region=not args.score_only)
            torch.cuda.synchronize()
        latency_ms = round((time.perf_counter() - t1) / args.latency_runs * 1e3, 3)
        np.random.set_state(state)

    train_line = None
    if args.train_steps > 0 and not args.score_only:
        # configs[3]'s training iteration on the same box, after the timed region (every rank takes part: the gradient
        # all-reduce is a collective)
        try:
            train_line = measure_train(args, rank, world, dev, args.train_steps, 5, args.train_batch)
        except Exception as exc:   # the headline line must survive a failure of this side measurement (all ranks raise alike)
            train_line = {"error": repr(exc)}

    if rank == 0:
        total_scenes = args.batch * args.steps * world
        agg = main_summary
        by_time = sorted(agg.items(), key=lambda kv: -kv[1][0])
        kernels = [{"op": k[0], "shape": k[1], "calls": c, "avg_ms": round(tot / c, 4), "total_ms": round(tot, 3)}
                   for k, (tot, c) in by_time]
        fam, roofline = roofline_of(agg, graph_accounting_steps or args.steps, args.batch,
                                    timer.critical if timer.critical_streams is not None else None)
        if roofline and graph_accounting_steps:
            roofline["accounting_note"] = ("the %d timed steps replayed hipGraphs (%d feature stages); launch durations are from %d "
                                           "extra steps outside the timed region with the same launches issued one by one"
                                           % (args.steps, graph_replays, graph_accounting_steps))
        # sa_chain_kernel skips the padded point tiles of small neighbourhoods (exact: a duplicate cannot move a maximum), so
        # the ALGORITHMIC flops of its launches (all 64 slots, what the reference multiplies) overstate what the matrix pipe
        # executes; every fraction quoted as a utilisation below is computed from the EXECUTED flops
        chain_exec = sa_chain_executed_share(score_net, pcs)
        launched_gflop = sum(f["units"] for f in fam.values() if f["bound"] == "mfma") / 1e9 / max(
            (graph_accounting_steps or args.steps) * args.batch, 1)
        chain_alg_gflop = (fam["sa_chain_kernel"]["units"] / 1e9 / max((graph_accounting_steps or args.steps) * args.batch, 1)
                           if "sa_chain_kernel" in fam else 0.0)
        share = chain_exec["executed_share_of_algorithmic_flops"] if chain_exec else 1.0
        executed_gflop = launched_gflop - chain_alg_gflop * (1.0 - share)
        if roofline and roofline["bound"] == "mfma":
            k_share = share if roofline["kernel"] == "sa_chain_kernel" else 1.0
            if roofline["kernel"] == "sa_chain_kernel" and chain_exec:
                roofline.update(chain_exec)
            roofline["frac_algorithmic"] = roofline["frac"]
            roofline["achieved_algorithmic"] = roofline["achieved"]
            roofline["frac"] = round(roofline["frac_algorithmic"] * k_share, 6)          # executed flops / event duration / peak
            roofline["achieved"] = round(roofline["achieved_algorithmic"] * k_share, 4)
            roofline["frac_executed"] = roofline["frac"]
            roofline["executed_flop_per_launch"] = round(roofline["algorithmic_units_per_launch"] * k_share)
            rp_ms, rp_src, rp_note = rocprof_launch(roofline["kernel"], args.batch, args.points)
            roofline["rocprof_avg_launch_ms"] = rp_ms
            roofline["rocprof_source"] = rp_src
            roofline["frac_rocprof"] = None
            if rp_note:
                roofline["rocprof_note"] = rp_note
            if rp_ms:
                # the same fraction from the committed trace's duration instead of this run's events (the trace is of
                # another run of this command on another box: durations differ by a few per cent)
                roofline["frac_rocprof"] = round(roofline["executed_flop_per_launch"] / (rp_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 6)
            # whole step: executed flops of ALL matrix kernels of a batch / ms_per_step / peak
            roofline["step_frac_executed"] = round(executed_gflop * args.batch / (dt / args.steps * 1e3)   # GFLOP / ms = TFLOP/s
                                                   / MFMA_F32_PEAK_TFLOPS, 6)
            roofline["frac_note"] = ("frac / achieved / frac_rocprof count EXECUTED flops (sa_chain_kernel skips padded point tiles: "
                                     "executed_share_of_algorithmic_flops); frac_algorithmic / achieved_algorithmic count SURVEY "
                                     "8(d)'s algorithmic flops and may exceed what a matrix pipe can do")
        if roofline:
            src = traffic_source()
            roofline["traffic_source"] = src
            roofline["traffic_source_stale"] = src["stale"]
        if roofline and getattr(pipe, "split_chain_tail", False):
            roofline["overlap_note"] = ("the previous batch's last chain kernel leaves its partial final round of row blocks (64 of "
                                        "1600 workgroup passes) on a side stream; it runs beside the first ~0.25 ms of this kernel's "
                                        "launches, whose duration includes that sharing")
        if roofline and args.mlp_streams > 1:
            roofline["concurrency"] = ("%d feature-stage streams: launches of this family overlap each other, so the launch "
                                       "duration above (what rocprofv3 shows too) includes time-sharing; roofline_exclusive "
                                       "is the same measurement with one feature-stage stream" % args.mlp_streams)
        res = {
            "metric": "scenes/sec (%s-pt ScoreNet+GRN fwd)" % _pts(args.points), "value": round(total_scenes / dt, 3),
            "unit": "scenes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[1]: ScoreNet forward" if args.score_only else
                                    "configs[2]: ScoreNet+GraspRegionNet+RefineNet forward") +
                                   ", %d-pt synthetic scenes, batch=%d per GPU, eval" % (args.points, args.batch),
                       "points": args.points, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": "scene-sharded x%d (no data-path collective)" % world,
                       "sampling_group_batches": args.fps_group or max(1, min(8, 64 // max(1, args.batch))),
                       # batches the pipeline pulled from its input before the first result could exist (the first sampling
                       # launch of the timed run); steady state: up to 2 x sampling_group_batches
                       "sampling_lookahead_batches": first_launch_batches,
                       # ... issued as this many launches of these sizes (pipeline.SPLIT_FIRST_LAUNCH: batch 1 alone on one
                       # sampling stream so that its features start 0.5 ms earlier, the rest on the other); null = one launch
                       "sampling_first_launch_split": list(first_launch_split) if first_launch_split else None,
                       # distinct batches the steps cycle through (seeds 1000 + (k*W + rank)*B ...)
                       "distinct_batches": distinct,
                       # automatic garbage collections of the host interpreter inside the timed region (all threads)
                       "host_gc": {"collections_by_generation": gc_log["n"], "ms": round(gc_log["ms"], 2)},
                       "region_head": region_calibration or "seeded only (refine stage skipped by the reference's own guard)",
                       # per timed step: grasps decoded by stage 2, closing boxes with > 5 points (= rows of the refine
                       # network), class-1 grasps the refine stage kept
                       "refine_stage": None if not refine_stats else {
                           "stage2_grasps_per_step": round(sum(r[0] for r in refine_stats) / len(refine_stats), 1),
                           "valid_crops_per_step": (None if any(r[1] is None for r in refine_stats) else
                                                    round(sum(r[1] for r in refine_stats) / len(refine_stats), 1)),
                           "class1_grasps_per_step": round(sum(r[2] for r in refine_stats) / len(refine_stats), 1),
                           "steps_with_refine": sum(1 for r in refine_stats if r[2] > 0 or (r[1] or 0) >= 2),
                           # launches of the refine network's own kernels in the timed region (their row count -- the
                           # valid crops of the step -- varies, so the per-shape list below holds one entry per count)
                           "kernel_calls": {"gather_max G64 (crop features + MaxPool1d(64))":
                                            sum(c for (n, m), (_, c) in agg.items() if n == "gather_max" and " G64 " in m),
                                            "mlp_layer K384 N1024 (conv_formal)":
                                            sum(c for (n, m), (_, c) in agg.items() if n == "mlp_layer" and " K384 N1024" in m)}},
                       # of the timed steps: feature stages enqueued when the previous batch's had already finished (the launching
                       # thread paced them, not the GPU), and feature stages whose geometry was still running when they were enqueued
                       "host_late_feature_stages": getattr(pipe, "host_late_feature_stages", None),
                       "geometry_pending_at_enqueue": getattr(pipe, "geometry_pending_at_enqueue", None),
                       "switches": args.set or None,
                       # multi-rank runs: what the process group really is (sharding.describe_collective); the forward path
                       # itself has no data-path collective -- the probe all-reduce is the training bucket's size
                       "collective": collective,
                       "hip_graphs": bool(graph_replays),
                       "scorenet_gflop_per_scene": SCORENET_GFLOP_PER_SCENE.get(args.points),
                       # flops of the launched matrix kernels per scene: "launched" after the algebraic restructuring (layer 1
                       # of levels 2-3 and of the propagation blocks evaluated per source point), "executed" also without the
                       # padded point tiles sa_chain_kernel skips
                       "launched_gflop_per_scene": round(launched_gflop, 2),
                       "executed_gflop_per_scene": round(executed_gflop, 2),
                       "level1_balls": chain_exec},
            "roofline": roofline,
            "roofline_exclusive": None,
            "latency_ms_single_scene": latency_ms,
            # the same pipeline on density-matched scenes (level-1 neighbourhoods like the reference's clouds), after the timed region
            "value_real_density": None if real_density is None else real_density["value"],
            # EXPERIMENT, never the headline: the same run with the level-1 block on the bf16 matrix pipe (fp32-faithful split products)
            "value_split_products": None if split_products is None else split_products["value"],
            "split_products": split_products,
            "real_density": real_density,
            "value_no_lookahead": None if no_lookahead is None else round(no_lookahead, 3),
            "value_no_lookahead_note": None if no_lookahead is None else (
                "%d steps after the timed region, --fps-group 1 --first-launch-groups 1 (one sampling launch per batch, "
                "look-ahead 3-4 batches)" % args.no_lookahead_steps),
            "mlp_tflops": round(SCORENET_GFLOP_PER_SCENE.get(args.points, 0.0) * total_scenes / world / dt / 1e3, 3),
            "kernels": kernels[:40],
            "grasps_last_step": int(out["next_grasp"].shape[0]) if "next_grasp" in out else None,
        }
        if exclusive is not None:
            _, r1 = roofline_of(exclusive[0], args.exclusive_steps, args.batch, timer.critical if timer.critical_streams is not None else None)
            if r1:
                r1["steps"] = args.exclusive_steps
                r1["scenes_per_s_in_this_mode"] = round(args.batch * args.exclusive_steps / exclusive[1], 1)
            res["roofline_exclusive"] = r1
        else:
            res.pop("roofline_exclusive")
        if train_line is not None and "error" in train_line:
            res["train"] = train_line
        elif train_line is not None:
            res["train"] = {"scenes_per_s": train_line["value"], "ms_per_step": train_line["ms_per_step"],
                            "steps": train_line["steps"], "warmup": train_line["warmup"],
                            "batch_per_gpu": args.train_batch, "roofline": train_line["roofline"],
                            "allreduce_ms": train_line["allreduce_ms"], "workload": train_line["config"]["workload"],
                            "ms_per_step_median": train_line["ms_per_step_median"], "iteration_ms": train_line["iteration_ms"],
                            "host_iteration_ms": train_line["host_iteration_ms"], "trunk_stream": train_line["trunk_stream"],
                            "graph_replays": train_line["graph_replays"], "readback_wait_ms": train_line["readback_wait_ms"],
                            "host_ms_per_step": train_line["host_ms_per_step"],
                            "host_blocked_ms_per_step": train_line["host_blocked_ms_per_step"],
                            "gpu_idle_ms_per_step": train_line["gpu_idle_ms_per_step"], "gpu_timeline": train_line["gpu_timeline"],
                            "distinct_batches": train_line["config"]["distinct_batches"],
                            "bucket_bytes": train_line["config"]["bucket_bytes"],
                            "parallelism": train_line["config"]["parallelism"],
                            "note": "measured after the timed region; `python bench.py --train` times it alone"}
        if world == 1 and args.cpu_scenes > 0:
            res["cpu_baseline"] = cpu_baseline(args, args.cpu_scenes, (score_net, region_net))
            res["parity"] = res["cpu_baseline"].pop("parity")
            res["speedup_vs_cpu_baseline"] = round(res["value"] / res["cpu_baseline"]["value"], 1)
        emit(res)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
