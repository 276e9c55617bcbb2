"""ForwardPipeline (six or seven HIP streams, several batches in flight; one or two feature-stage streams; the level-1 sampling of
consecutive batches grouped into one launch or not) must return exactly what the sequential forward returns for the same batches
and numpy seed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("mlp_streams,fps_group", [(1, 0), (2, 0), (1, 2), (1, 1)])
def test_pipeline_equals_sequential(mlp_streams, fps_group):
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    score_net, region_net = pipeline.build_models(DEV)
    batches = [synthetic.make_batch(3000 + 10 * i, 2, 6144, device=DEV) for i in range(5)]   # 5: the last sampling group is partial
    synthetic.calibrate_score_head(score_net, batches[0])
    np.random.seed(77)
    want = [pipeline.forward_scenes(score_net, region_net, pc) for pc in batches]
    torch.cuda.synchronize()
    np.random.seed(77)
    pipe = pipeline.ForwardPipeline(score_net, region_net, mlp_streams=mlp_streams, fps_group=fps_group)
    got = list(pipe.run(iter(batches)))
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert torch.equal(g["score"], w["score"])
        assert torch.equal(g["all_feature"], w["all_feature"])
        for key in ("center_pc_index", "pc_group_index", "pc_group_more_index"):
            assert torch.equal(g[key], w[key]), key
        # every stage (incl. the grasp-region / refine heads) runs on this repo's deterministic kernels
        assert torch.equal(g["next_grasp"], w["next_grasp"])
        if w["final_mask"] is None:
            assert g["final_mask"] is None
        else:
            assert torch.equal(g["final_mask"], w["final_mask"])
            assert torch.equal(g["select_grasp_class"], w["select_grasp_class"])


def test_plan_then_forward_equals_forward():
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    score_net, _ = pipeline.build_models(DEV)
    pc = synthetic.make_batch(3100, 2, 6144, device=DEV)
    with torch.no_grad():
        f0, s0, _ = score_net(pc)
        plan = score_net.plan(pc)
        f1, s1, _ = score_net(pc, plan=plan)
    assert torch.equal(f0, f1) and torch.equal(s0, s1)


def _same_outputs(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        for key in ("score", "all_feature", "center_pc_index", "pc_group_index", "pc_group_more_index", "next_grasp"):
            assert torch.equal(g[key], w[key]), key


def test_stage_graphs_replay_the_same_bits_and_follow_weight_and_shape_changes():
    """ForwardPipeline(graphs=True): the geometry and feature stages replayed as hipGraphs (pipeline._StageGraphs) give the
    bits of the launch-by-launch pipeline; results handed out stay valid while later batches reuse the slot; a second run
    reuses the graphs; an in-place weight update or another batch shape re-captures."""
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    score_net, region_net = pipeline.build_models(DEV)
    batches = [synthetic.make_batch(3300 + 10 * i, 2, 6144, device=DEV) for i in range(7)]
    synthetic.calibrate_score_head(score_net, batches[0])

    def run(pipe, items, seed=5):
        np.random.seed(seed)
        out = list(pipe.run(iter(items)))
        torch.cuda.synchronize()
        return out

    eager = pipeline.ForwardPipeline(score_net, region_net, graphs=False)
    want = run(eager, batches)
    assert eager.graph_replays == 0
    pipe = pipeline.ForwardPipeline(score_net, region_net, graphs=True)
    got = run(pipe, batches)
    assert pipe.graph_replays == len(batches)        # the capture happens in front of the first batch's geometry
    _same_outputs(got, want)                         # (all 7 results compared AFTER the run: slots were reused 2-3 times)
    graphs = pipe._stage_graphs
    got = run(pipe, batches[:3])
    assert pipe._stage_graphs is graphs and pipe.graph_replays == len(batches) + 3
    _same_outputs(got, want[:3])
    # in-place weight update (what an optimizer step or load_state_dict does): stale packed weights must not be replayed
    with torch.no_grad():
        score_net.extrat_featurePN2.sa_modules[1].mlp[1].conv.weight.mul_(1.01)
    want2 = run(eager, batches[:3])
    got2 = run(pipe, batches[:3])
    assert pipe._stage_graphs is not graphs
    assert not torch.equal(want2[0]["score"], want[0]["score"])
    _same_outputs(got2, want2)
    # another batch shape in the same run
    mixed = [batches[0], batches[1][:1].contiguous(), batches[2][:1].contiguous(), batches[3]]
    _same_outputs(run(pipe, mixed), run(eager, mixed))


def test_stage_graph_capture_survives_stale_graphs_in_reference_cycles():
    """Pipelines (and their captured graphs) that ended up in reference cycles are destroyed by the cyclic collector at an
    arbitrary allocation -- if that were in the middle of the next pipeline's stream capture, the graph destructor's
    hipGraphDestroy would be refused and the exception from a destructor would end the process.  _StageGraphs collects
    first and keeps the collector off while it captures; here the collector is made eager to provoke the overlap."""
    import gc
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    score_net, region_net = pipeline.build_models(DEV)
    batches = [synthetic.make_batch(3500 + i, 1, 6144, device=DEV) for i in range(3)]
    old = gc.get_threshold()
    gc.set_threshold(20, 2, 2)
    try:
        outs = []
        for trial in range(5):
            pipe = pipeline.ForwardPipeline(score_net, region_net, with_region=False, graphs=True)
            pipe.myself = pipe                                   # a cycle: only the cyclic collector can free pipe and its graphs
            outs.append([o["score"].clone() for o in pipe.run(iter(batches))])
            torch.cuda.synchronize()
            assert pipe.graph_replays == len(batches)
    finally:
        gc.set_threshold(*old)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
