"""The forward hot path end to end: ScoreNet -> region grouping -> grasp-region + refine
(the step of test.py:134-141 / train.py:358-367 without losses), as one callable used by
bench.py, smoke() and the tests."""
import contextlib
import io

import torch

from . import synthetic
from .get_regiondataset import get_grasp_allobj
from .gripper_region_network import GripperRegionNetwork
from .score_network import ScoreNetwork

# hyper-parameters hard-coded by the reference's train.py:70-90
CENTER_NUM, SCORE_THRE = 64, 0.5
GROUP_NUM, R_TIME_GROUP = 256, 0.1
GROUP_NUM_MORE, R_TIME_GROUP_MORE = 1024, 0.8
WIDTH, HEIGHT, DEPTH = 0.08, 0.01, 0.06
GRIPPER_NUM, GRASP_SCORE_THRESHOLD, REG_CHANNEL = 64, 0.5, 10

PARAMS = [CENTER_NUM, SCORE_THRE, GROUP_NUM, R_TIME_GROUP, GROUP_NUM_MORE, R_TIME_GROUP_MORE, WIDTH, HEIGHT, DEPTH]
GRIPPER_PARAMS = [WIDTH, HEIGHT, DEPTH]


def build_models(device, score_seed=7, region_seed=11):
    """ScoreNetwork + GripperRegionNetwork in eval mode with seeded (numpy) weights."""
    score_net = ScoreNetwork(training=True)
    score_net.load_state_dict(synthetic.seeded_state_dict(score_net, score_seed))
    region_net = GripperRegionNetwork(training=True, group_num=GROUP_NUM, gripper_num=GRIPPER_NUM,
                                      grasp_score_threshold=GRASP_SCORE_THRESHOLD, radius=DEPTH,
                                      reg_channel=REG_CHANNEL)
    region_net.load_state_dict(synthetic.seeded_state_dict(region_net, region_seed))
    return score_net.to(device).eval(), region_net.to(device).eval()


def forward_scenes(score_net, region_net, pc, with_region=True):
    """pc (B,N,6) -> dict(all_feature, score, centres, next_grasp, select_grasp_class, ...)."""
    with torch.no_grad():
        all_feature, score, _ = score_net(pc)
        out = {"all_feature": all_feature, "score": score}
        if not with_region:
            return out
        (center_pc, center_idx, g_idx, g, gm_idx, gm, _) = get_grasp_allobj(pc, score, PARAMS, [])
        with contextlib.redirect_stdout(io.StringIO()):
            res = region_net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, all_feature, GRIPPER_PARAMS, None, [])
    out.update(center_pc_index=center_idx, pc_group_index=g_idx, pc_group_more_index=gm_idx, next_grasp=res[0],
               keep_per_scene=res[1], true_mask=res[2], select_grasp_class=res[6], select_grasp_score=res[7],
               final_mask=res[11])
    return out


SINGLE_CU_POINTS = 25600   # largest scene whose furthest point sampling runs on ONE workgroup (csrc/geometry.hip)
NUM_CUS = 256              # MI355X


def sampling_workgroups_per_scene(num_points):
    """Workgroups (= whole CUs) one scene's level-1 sampling holds: 1 up to 25 600 points, else 2-4 COOPERATING workgroups
    (fps_multi_kernel) that exchange a word per round and therefore must all be resident at the same time."""
    return max(1, -(-int(num_points) // SINGLE_CU_POINTS))


# ForwardPipeline default for per-level readiness events between sampling, geometry and features.  On: the first batch of a run
# starts its level-1 block ~2.5 ms earlier (+0.3-1 % over a 20-step run) -- beside the level-2/3 sampling of every batch of the
# first launch, which holds up to 160 CUs for 0.8 ms: that one launch of the dominant kernel then takes ~3 ms instead of 1.9 and
# the run's average launch duration (what `roofline` is computed from) reads 3 % longer.  Off by default: per-kernel durations
# stay a property of the kernel; a latency-minded caller switches it on.
LEVEL_EVENTS = False
# ... but for the FIRST batch of a run only: nothing else is on the chip while its geometry is computed, so its level-1 block
# may start as soon as level 1's ball query is there (levels 2-3 and the 3-NN searches then run beside it)
FIRST_BATCH_LEVEL_EVENTS = True
# The first sampling launch of a run as TWO launches on the two sampling streams: the first batch alone, the rest of the look-ahead
# beside it -- the first batch's centroids are there when ITS 8 workgroups are done, not when all 160 are (14 alternating 20-step runs on one
# box: 7.943 -> 7.913 ms per step, medians 7.934 -> 7.913; outputs do not depend on how scenes are grouped into launches)
SPLIT_FIRST_LAUNCH = True
GRAPH_MAX_POINTS = 4 * 25600   # graphs="auto": batches of at most this many points replay hipGraphs (launch-bound shapes)


class _StageGraphs:
    """hipGraph replays of the geometry stage and the feature stage for ONE batch shape.

    A batch of 1-4 scenes is launch-bound: its geometry + feature stages are ~75 launches (ctypes calls, torch allocations
    and views: 0.9-1.3 ms of host time) for 1.7 ms of device time at one scene, and the interpreter is shared with the
    host-paced region stage.  Both stages depend on the batch only through its values -- shapes, launch grids and every
    address are fixed by (B, N) -- so each is captured once as a hipGraph over static buffers and replayed per batch:
    65 us of host time for both (scripts/graph_probe.py), the same kernels in the same order, bit-identical results
    (tests/test_gpu_pipeline.py).  ``slots`` sets of buffers rotate so that the geometry of batch i+1 can be written while
    the features of batch i are read; the stage outputs are copied out of the slot (26 MB per scene, ~10 us) so that
    results handed to the caller are never overwritten.  The level-1..3 sampling stays outside (grouped launches of
    varying size).  Captured addresses include the packed weights: ``signature`` (parameter versions + the fused module's
    switches) is compared at the start of every ``run`` and the graphs are dropped when it changed."""

    def __init__(self, pipe, pc, ctr, signature, slots=3):
        from . import fused
        self.signature = signature
        self.key = _graph_key(pc)
        self.slots = []
        self.next = 0
        dev = pc.device
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            # one eager pass first: every lazily built cache (interpolation tables, grid workspaces, LDS opt-ins) must exist
            # BEFORE a capture, or it would be allocated from a graph's private pool and filled by replays only
            plan = pipe._plan(pc, ctr)
            pipe.score_net(pc, plan=plan)
            cur.synchronize()
            self._capture_slots(pipe, pc, ctr, slots, fused)
            torch.cuda.synchronize(dev)

    def _capture_slots(self, pipe, pc, ctr, slots, fused):
        # No destructor of a stale CUDAGraph may run while a stream captures (hipGraphDestroy is "not permitted when stream is
        # capturing"; raised from a destructor it terminates the process): garbage graphs -- an earlier pipeline caught in a
        # reference cycle -- are collected NOW, and the cyclic collector stays off until the last capture has ended.
        import gc
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            for _ in range(slots):
                slot = {"pc": pc.clone(), "ctr": [c.clone() for c in ctr]}
                slot["g_geo"], slot["g_feat"] = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # thread_local: the region worker thread may allocate or synchronise while this thread captures
                with torch.cuda.graph(slot["g_geo"], stream=pipe.s_geo, capture_error_mode="thread_local"):
                    slot["plan"] = pipe._plan(slot["pc"], slot["ctr"])
                old_sink, fused.TAIL_SINK = fused.TAIL_SINK, None    # no side-stream tail inside a graph
                try:
                    with torch.cuda.graph(slot["g_feat"], stream=pipe.s_mlp, capture_error_mode="thread_local"):
                        slot["all_feature"], slot["score"], _ = pipe.score_net(slot["pc"], plan=slot["plan"])
                finally:
                    fused.TAIL_SINK = old_sink
                slot["free"] = None     # event: the slot's last feature replay and the copies of its outputs are done
                self.slots.append(slot)
        finally:
            if gc_was_enabled:
                gc.enable()

    def take(self):
        slot = self.slots[self.next % len(self.slots)]
        self.next += 1
        return slot


_stream_pools = {}


def reserve_streams(device, fps_streams=2, mlp_streams=1):
    """The HIP streams of a ``ForwardPipeline`` on ``device`` -- created once per (device, counts) and shared by every
    pipeline built afterwards -- as {"fps": [...], "mlp": [...], "geo": stream, "reg": stream}.

    The HIP runtime binds a stream to a hardware queue when its handle is first used, in order of first use, and the
    binding matters: with the feature stream bound FIRST (e.g. a caller reading ``s_mlp.cuda_stream`` before the first
    ``run``) a step takes 9.1 ms instead of 8.4 (scripts/stream_order_probe.py); with RCCL's streams bound before these --
    ``init_process_group("nccl")`` ahead of the first pipeline -- 8.5 instead of 7.75 (scripts/ablate/group_overhead_ab.sh:
    every chain kernel 3 % slower, the plain layers 30 %, the side streams' small kernels queueing behind one another).  So
    they are bound HERE, in the order the first ``run`` would use them -- sampling, features, geometry, region -- and
    ``sharding.init`` calls this BEFORE it creates the process group."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, max(1, int(fps_streams)), max(1, int(mlp_streams)))
    pool = _stream_pools.get(key)
    if pool is None:
        # Priorities: FPS and the region stage are chains of small / single-CU kernels (the latter separated by host
        # syncs) -- they must not queue behind the big MLP launches.
        pool = {"fps": [torch.cuda.Stream(dev, priority=-1) for _ in range(key[1])]}
        pool["geo"] = torch.cuda.Stream(dev, priority=-1)
        pool["mlp"] = [torch.cuda.Stream(dev, priority=0) for _ in range(key[2])]
        pool["reg"] = torch.cuda.Stream(dev, priority=-1)
        for st in tuple(pool["fps"]) + tuple(pool["mlp"]) + (pool["geo"], pool["reg"]):
            _ = st.cuda_stream
        _stream_pools[key] = pool
    return pool


def _graph_key(pc):
    return (tuple(pc.shape), tuple(pc.stride()), pc.dtype, pc.device.index)


def _graph_signature(score_net):
    from . import fused
    switches = tuple(sorted((k, v) for k, v in vars(fused).items() if k.isupper() and isinstance(v, (bool, int, float))))
    return (fused._signature(score_net), switches, score_net.training)


class ForwardPipeline:
    """Software pipeline over a stream of scene batches (inference).

    Scenes are independent, and inside one batch the sampling / grouping geometry depends on xyz
    only.  Level-1 furthest point sampling is a latency-bound chain of dependent argmax rounds (~880 rounds of
    ~6 picks since round 3) that keeps one CU per scene busy for ~4 ms (10 ms in rounds 1-2), while the shared-MLP contraction wants the other ~250
    CUs and region grouping needs the host for numpy's RNG.  So four stages run concurrently on five
    (or more) HIP streams, each on a different batch:

        s_fps : sample(batches i+2 ..) level-1 FPS of the NEXT GROUP of batches (up to 64 scenes) in one launch, groups
                                      alternating between two streams: a launch is a ~4-5 ms latency chain on ONE CU per
                                      scene, and what it costs the matrix kernels does not grow with its size up to one or
                                      two CUs per shader engine (see _sample_group), so it pays to run it rarely
        s_geo : geometry(batch i+1)   FPS levels 2-3, ball query x3, 3-NN x3
        s_mlp : features(batch i)     gather / MFMA shared-MLP / pool / head    (matrix cores; with ``mlp_streams=2``
                                      consecutive batches alternate between two such streams)
        s_reg : region(batch i-1)     radius grouping, host RNG draws, GRN + refine heads

    Results are identical to running ``forward_scenes`` batch by batch (same kernels, same numpy
    RNG call order: region stages execute in batch order on the host thread).
    """

    def __init__(self, score_net, region_net, with_region=True, fps_streams=2, mlp_streams=1, fps_group=0,
                 first_launch_groups=1, geometry_ahead=1, graphs="auto"):
        """``fps_streams``: level-1 sampling launches in flight (each ~4-5 ms on one CU per scene, whatever the batch
        size).  ``fps_group``: consecutive batches whose level-1 sampling shares one launch (0: as many as give 64 scenes,
        at most 8): +6 % at 8 scenes per batch, +12-15 % at 1 and 4, at the price of reading that many batches ahead.
        ``first_launch_groups``: the FIRST sampling launch of a ``run`` finds the chip idle (nothing can start before its
        result) and may take this many groups at once (bench.py passes 4: all of a short run's batches are then sampled in
        one launch); the library default is 1.  INPUT-LATENCY CONTRACT: ``run`` pulls up to
        ``first_launch_groups x group`` batches from its iterator before it yields the first result and stays up to
        ``2 x group`` batches ahead afterwards -- with a live source (camera, data loader) use ``fps_group=1``.
        ``mlp_streams``: feature stages (consecutive batches) that may overlap.  One batch's ~30 MFMA launches leave the
        chip partly idle at every kernel tail and in the small layers (P <= 40 960 rows); a second stream fills those
        holes with the next batch's kernels: 9.70 -> 9.14 ms per batch of 8 (824 -> 875 scenes/s) with
        ``mlp_streams=2``; 3 measured slower.  The default stays 1 because overlapping launches time-share the chip:
        a launch's duration (the quantity the roofline accounting and every profile under profiles/ is built on) then
        depends on what the other stream happens to run, and a profiler perturbs exactly that.
        ``graphs``: replay the geometry and feature stages as hipGraphs (see _StageGraphs): ``"auto"`` for batches of at most
        GRAPH_MAX_POINTS points (launch-bound: +40 % at one scene per batch, steadier at 3-4), True / False to force."""
        self.score_net, self.region_net, self.with_region = score_net, region_net, with_region
        self.graphs = graphs
        self._stage_graphs = None         # _StageGraphs of the shape the last batches had
        self.graph_replays = 0            # feature stages served by a replay (bench.py reports it)
        self._graph_sig = None
        self.fps_group = int(fps_group)   # batches whose level-1 sampling shares one launch; 0 = as many as give 64 scenes (<= 8)
        self.first_launch_groups = max(1, int(first_launch_groups))
        self.geometry_ahead = max(1, int(geometry_ahead))
        self.level_events = LEVEL_EVENTS  # geometry and features synchronise per network level (see _geometry)
        self.split_chain_tail = True      # see _features   # batches whose ball-query / 3-NN geometry runs ahead of the features
        self.first_launch_batches = None  # batches the first sampling look-ahead of the last ``run`` took (bench.py reports it)
        self.first_launch_split = None    # ... and, when it was issued as two launches (SPLIT_FIRST_LAUNCH), their sizes
        self._one_sampling_stream = False
        dev = next(score_net.parameters()).device
        self.device = dev
        # Priorities: FPS and the region stage are chains of small / single-CU kernels (the latter
        # separated by host syncs) -- they must not queue behind the big MLP launches.
        pool = reserve_streams(dev, fps_streams, mlp_streams)
        self.s_fps, self.s_geo, self.s_mlps, self.s_reg = pool["fps"], pool["geo"], pool["mlp"], pool["reg"]
        self._n_sampled = 0
        self.s_mlp = self.s_mlps[0]
        self._n_featured = 0

    # -- stages -------------------------------------------------------------------------------
    def _sample_group(self, pcs, first=False):
        """Level-1 sampling of several consecutive batches in ONE launch -> one item per batch.

        A sampling workgroup owns a whole CU for ~4-5 ms (10 ms when this was measured).  The hardware deals the workgroups of every launch to the XCDs and
        their shader engines in a fixed rotation, in order, so an engine that has lost a CU paces all the others: 8 held
        CUs (one per XCD) cost the matrix kernels as much as 32 (one per engine) -- 12.5 %, measured (DESIGN.md par. 10).
        So the sampling of up to 64 scenes (two per engine) goes into one launch: the same cost while it runs, but it runs
        an eighth of the time."""
        # scenes beyond one CU's register file sample on cooperating workgroups that must ALL be resident: such launches
        # stay on ONE stream (stream order = never two of them in flight), see ``run``
        stream = self.s_fps[0 if self._one_sampling_stream else self._n_sampled % len(self.s_fps)]
        self._n_sampled += 1
        with torch.cuda.stream(stream), torch.no_grad():
            # all three sampling levels: the level-2 / level-3 launches hold a CU per scene too (1.3 + 0.5 ms)
            big = pcs[0] if len(pcs) == 1 else torch.cat(pcs, 0)
            level1_done = None
            first = first and FIRST_BATCH_LEVEL_EVENTS and not self.level_events
            if (self.level_events or first) and "_sample" not in self.__dict__:
                marks = {}

                def after_level(i):
                    if i == 0:
                        marks[0] = torch.cuda.Event()
                        marks[0].record(stream)
                ctr = self.score_net.sample_levels(big, after_level)
                level1_done = marks.get(0)
            else:
                ctr = self._sample(big)
            done = torch.cuda.Event()
            done.record(stream)
        for c in ctr:
            c.record_stream(self.s_geo)
            for m in self.s_mlps:
                c.record_stream(m)
        items, at = [], 0
        for pc in pcs:
            items.append({"pc": pc, "ctr": [c[at:at + pc.shape[0]] for c in ctr], "fps_done": done,
                          "fps1_done": level1_done if (not first or not items) else None})   # first: the run's first batch only
            at += pc.shape[0]
        return items

    def _sample(self, big):
        """Sampled centroid indices of every level for the concatenated scenes (overridden by timing harnesses under
        scripts/ only)."""
        return self.score_net.sample_levels(big)

    def _plan(self, pc, ctr):
        return self.score_net.plan(pc, ctr)

    def _graphs_for(self, pc):
        """The shape's _StageGraphs when this batch should replay graphs, else None."""
        want = self.graphs
        if want == "auto":
            want = pc.shape[0] * pc.shape[1] <= GRAPH_MAX_POINTS
        from . import fused
        if not want or not fused.ENABLED or len(self.s_mlps) != 1 or self.score_net.training:
            return None
        g = self._stage_graphs
        return g if g is not None and g.key == _graph_key(pc) else None

    def _capture(self, item):
        """Build the graphs for this batch's shape (once per shape and weight version; needs the batch's sampled
        centroids, so it runs when the first batch reaches the geometry stage -- a device-wide synchronisation)."""
        item["fps_done"].synchronize()
        self._stage_graphs = None
        if self._graph_sig is None:
            self._graph_sig = _graph_signature(self.score_net)
        self._stage_graphs = _StageGraphs(self, item["pc"], item["ctr"], self._graph_sig, slots=self.geometry_ahead + 2)

    def _geometry_replay(self, item, graphs):
        slot = graphs.take()
        with torch.cuda.stream(self.s_geo), torch.no_grad():
            self.s_geo.wait_event(item["fps_done"])
            if slot["free"] is not None:
                self.s_geo.wait_event(slot["free"])     # the features that last read this slot's plan and points
            slot["pc"].copy_(item["pc"], non_blocking=True)
            for dst, src in zip(slot["ctr"], item["ctr"]):
                dst.copy_(src, non_blocking=True)
            slot["g_geo"].replay()
            done = torch.cuda.Event()
            done.record(self.s_geo)
        item.update(slot=slot, geo_done=done)
        return item

    def _features_replay(self, item):
        slot = item.pop("slot")
        s_mlp = self.s_mlp
        self._n_featured += 1
        self.graph_replays += 1
        with torch.cuda.stream(s_mlp), torch.no_grad():
            s_mlp.wait_event(item["geo_done"])
            slot["g_feat"].replay()
            all_feature, score = slot["all_feature"].clone(), slot["score"].clone()
            done = torch.cuda.Event()
            done.record(s_mlp)
            slot["free"] = done
        all_feature.record_stream(self.s_reg)
        score.record_stream(self.s_reg)
        item.update(all_feature=all_feature, score=score, mlp_done=done, tail_done=[])
        item.pop("ctr")
        return item

    def _geometry(self, item):
        from . import fused
        graphs = self._graphs_for(item["pc"])
        if graphs is None and self._wants_graphs(item["pc"]):
            self._capture(item)
            graphs = self._graphs_for(item["pc"])
        if graphs is not None:
            return self._geometry_replay(item, graphs)
        with torch.cuda.stream(self.s_geo), torch.no_grad():
            if item.get("fps1_done") is not None and "_plan" not in self.__dict__:
                # level by level: level 1's ball query needs level 1's sampling only, and every level's geometry carries its own
                # completion event (PointNet2Seg.forward waits per level) -- the first batch of a run starts its level-1 block
                # ~2.5 ms earlier (levels 2-3 sampling, their ball queries and the three 3-NN searches run beside it)
                s_geo = self.s_geo

                def on_level(kind, i, geo):
                    if geo is None:
                        s_geo.wait_event(item["fps1_done"] if (kind, i) == ("sa", 0) else item["fps_done"])
                    else:
                        geo["ready"] = torch.cuda.Event()
                        geo["ready"].record(s_geo)
                plan = self.score_net.plan(item["pc"], item["ctr"], on_level)
                done = plan["sa"][0]["ready"]
            else:
                self.s_geo.wait_event(item["fps_done"])
                plan = self._plan(item["pc"], item["ctr"])
                done = torch.cuda.Event()
                done.record(self.s_geo)
        for t in fused.plan_tensors(plan):
            for m in self.s_mlps:
                t.record_stream(m)
        item.update(plan=plan, geo_done=done)
        return item

    def _wants_graphs(self, pc):
        want = self.graphs
        if want == "auto":
            want = pc.shape[0] * pc.shape[1] <= GRAPH_MAX_POINTS
        from . import fused
        return bool(want) and fused.ENABLED and len(self.s_mlps) == 1 and not self.score_net.training and pc.is_cuda

    def _features(self, item):
        if "slot" in item:
            return self._features_replay(item)
        s_mlp = self.s_mlps[self._n_featured % len(self.s_mlps)]
        self._n_featured += 1
        from . import fused
        # accounting only (bench.py: config.host_late_feature_stages): was the feature stream already idle -- the previous
        # batch's feature stage finished -- when this batch's stage is enqueued?  Then the launching thread, not the GPU, paced it.
        prev = self.__dict__.get("_last_mlp_done")
        if prev is not None and prev.query():
            self.host_late_feature_stages = self.__dict__.get("host_late_feature_stages", 0) + 1
        geo_pending = not item["geo_done"].query()
        if geo_pending:
            self.geometry_pending_at_enqueue = self.__dict__.get("geometry_pending_at_enqueue", 0) + 1
        with torch.cuda.stream(s_mlp), torch.no_grad():
            s_mlp.wait_event(item["geo_done"])
            # the last, partial round of the final chain kernel runs on a side stream beside the NEXT batch's first kernels
            # (fused.TAIL_SINK); whoever reads this batch's feature / scores waits for that event too
            fused.TAIL_SINK = tails = [] if self.split_chain_tail else None
            try:
                all_feature, score, _ = self.score_net(item["pc"], plan=item["plan"])
            finally:
                fused.TAIL_SINK = None
            done = torch.cuda.Event()
            done.record(s_mlp)
            self._last_mlp_done = done
            item["tail_done"] = list(tails or ())
        all_feature.record_stream(self.s_reg)
        score.record_stream(self.s_reg)
        item.update(all_feature=all_feature, score=score, mlp_done=done)
        item.pop("plan")
        item.pop("ctr")
        return item

    def _region(self, item):
        pc, all_feature, score = item["pc"], item["all_feature"], item["score"]
        out = {"all_feature": all_feature, "score": score}
        with torch.cuda.stream(self.s_reg), torch.no_grad():
            self.s_reg.wait_event(item["mlp_done"])
            for ev in item.get("tail_done", ()):
                self.s_reg.wait_event(ev)
            if self.with_region:
                (center_pc, center_idx, g_idx, g, gm_idx, gm, _) = get_grasp_allobj(pc, score, PARAMS, [])
                with contextlib.redirect_stdout(io.StringIO()):
                    res = self.region_net(g, gm, g_idx, gm_idx, center_pc, center_idx, pc, all_feature,
                                          GRIPPER_PARAMS, None, [])
                out.update(center_pc_index=center_idx, pc_group_index=g_idx, pc_group_more_index=gm_idx,
                           next_grasp=res[0], keep_per_scene=res[1], true_mask=res[2], select_grasp_class=res[6],
                           select_grasp_score=res[7], final_mask=res[11],
                           valid_crops=getattr(self.region_net, "last_valid_crops", None))
            done = torch.cuda.Event()
            done.record(self.s_reg)
        if not self.with_region:
            # nothing here blocks the host, so without this the driver thread would enqueue every remaining batch at
            # once; a bounded look-ahead (max_pending_regions batches) is what the full pipeline runs with
            item["mlp_done"].synchronize()
            for ev in item.get("tail_done", ()):
                ev.synchronize()
        out["done"] = done
        return out

    # -- driver -------------------------------------------------------------------------------
    def run(self, batches, max_pending_regions=3):
        """batches: iterable of (B,N,6) GPU tensors (already resident).  Yields one result dict per
        batch, in order.  The caller must ``result['done'].synchronize()`` (or synchronise the
        device) before reading results on another stream.

        The region stage blocks its host thread on several device->host syncs (candidate counts for
        the numpy draws); it therefore runs on ONE worker thread, in batch order (so numpy's global
        RNG is consumed exactly as in the sequential forward), while this thread keeps the three
        asynchronous stages fed.  At most ``max_pending_regions`` batches wait for their region stage.
        """
        import queue
        import threading

        from . import fused
        cur = torch.cuda.current_stream(self.device)
        with torch.no_grad():
            fused.prepack(self.score_net, self.region_net)   # packed-weight caches: built on `cur`, before the streams fork
        self._graph_sig = _graph_signature(self.score_net)
        if self._stage_graphs is not None and self._stage_graphs.signature != self._graph_sig:
            self._stage_graphs = None                        # weights or switches changed: the captured addresses are stale
        streams = tuple(self.s_fps) + tuple(self.s_mlps) + (self.s_geo, self.s_reg)
        for s in streams:
            s.wait_stream(cur)

        todo, done = queue.Queue(), queue.Queue()

        from . import np_random

        def region_worker():
            torch.cuda.set_device(self.device)
            # numpy's generator state stays on the device between the region stages of a run (they are its only users, in
            # batch order); it is handed back to np.random once, when the worker ends
            with np_random.deferred():
                while True:
                    item = todo.get()
                    if item is None:
                        return
                    try:
                        done.put(self._region(item))
                    except BaseException as exc:  # surface the failure in the consumer thread
                        done.put(exc)
                        return

        worker = threading.Thread(target=region_worker, name="regnet-region-stage", daemon=True)
        worker.start()
        pending = 0

        def collect(block):
            res = done.get() if block else done.get_nowait()
            if isinstance(res, BaseException):
                raise res
            return res

        try:
            import collections
            sampled = collections.deque()   # batches whose level-1 FPS is enqueued
            geo_q = collections.deque()     # batches whose remaining geometry is enqueued, oldest first
            it = iter(batches)
            exhausted = False
            group = self.fps_group
            first_launch, first_want = True, 0
            first_features = True
            self.first_launch_batches = None
            while True:
                # keep the sampling ahead: a new group launch when at most one group's worth of sampled batches is left (a
                # launch is a latency chain of about half a step (longer than one in rounds 1-2): with one batch per launch two must be in flight).
                # Cooperative launches (scenes beyond 25 600 points) share one stream, so two of THEM never overlap.
                while not exhausted and (first_want <= 0 or len(sampled) <= group):
                    pcs = []
                    # the very first launch finds the chip idle (nothing can run before its result): it may take
                    # ``first_launch_groups`` groups
                    want = first_want if first_launch else group
                    while first_want <= 0 or len(pcs) < want:
                        try:
                            pcs.append(next(it))
                        except StopIteration:
                            exhausted = True
                            break
                        if first_want <= 0:
                            # scenes one launch may hold.  A scene beyond 25 600 points samples on G = 2-4 COOPERATING
                            # workgroups (csrc/geometry.hip: fps_multi_kernel) that spin on each other's words: every
                            # workgroup of every such launch IN FLIGHT must be resident, or two half-resident launches
                            # wait for each other's CUs for ever.  So with G > 1: one sampling stream (launches are
                            # serialised by stream order) and a launch leaves room for the only other cooperative
                            # launch that can run beside it -- the region stage's centre picker over the positives of
                            # ONE batch (at most B0 x G workgroups on its own stream).
                            B0, N0 = max(1, pcs[0].shape[0]), pcs[0].shape[1]
                            G = sampling_workgroups_per_scene(N0)
                            self._one_sampling_stream = G > 1
                            cap = NUM_CUS if G == 1 else max(B0, (NUM_CUS - B0 * G) // G)
                            if group <= 0:
                                group = max(1, min(8, min(64, cap) // B0))
                            group = max(1, min(group, max(1, cap // B0)))
                            first_want = max(group, min(self.first_launch_groups * group, cap // B0))
                            want = first_want if first_launch else group
                    if pcs:
                        if first_launch:
                            self.first_launch_batches = len(pcs)
                            self.first_launch_split = None
                        if (first_launch and SPLIT_FIRST_LAUNCH and len(pcs) > 1 and len(self.s_fps) > 1
                                and not self._one_sampling_stream):
                            # (two launches on two streams: batch 1 alone, then the rest of the look-ahead)
                            self.first_launch_split = (1, len(pcs) - 1)
                            sampled.extend(self._sample_group(pcs[:1], first=True))
                            sampled.extend(self._sample_group(pcs[1:], first=False))
                        else:
                            sampled.extend(self._sample_group(pcs, first=first_launch))
                        first_launch = False
                if not sampled and not geo_q:
                    break
                # enqueue the asynchronous stages, deepest look-ahead first.  ``geometry_ahead`` batches stay queued BEHIND
                # the one whose features are enqueued now (1 = one batch of geometry in flight beside the feature stage;
                # 2-3 measured the same step time: the feature stream never waits for geometry, scripts/mlp_idle_probe.py)
                while sampled and len(geo_q) < self.geometry_ahead + 1:
                    geo_q.append(self._geometry(sampled.popleft()))
                    if first_features:
                        break            # start of a run: the first batch's features should not queue behind a second geometry
                if geo_q and (first_features or len(geo_q) > self.geometry_ahead or not sampled):
                    todo.put(self._features(geo_q.popleft()))
                    pending += 1
                    first_features = False
                while pending > max_pending_regions:      # back-pressure: wait for the oldest region stage
                    yield collect(True)
                    pending -= 1
                while pending:                             # hand out whatever is already finished
                    try:
                        res = collect(False)
                    except queue.Empty:
                        break
                    yield res
                    pending -= 1
            while pending:
                yield collect(True)
                pending -= 1
        finally:
            todo.put(None)
            worker.join()
        for s in streams:
            cur.wait_stream(s)
        from . import pn2_ext
        pn2_ext.raise_if_fps_failed()         # (only holds flags when cooperative sampling launches were issued)
        if self.with_region:
            from . import region_ops
            region_ops.raise_if_out_of_range()
