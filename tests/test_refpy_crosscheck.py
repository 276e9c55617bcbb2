"""``-m refpy``: the host-side mirror against the reference's LIVE Python graph (authoring container only -- the tests
skip where /root/reference is absent, i.e. on the GPU box).  The committed fixtures (tests/golden/*) pin fixed seeds; this
cross-check draws ANOTHER seed every time the fixtures are regenerated -- the seed below is not one any fixture used --
so a mirror that matched the fixtures by accident (or a fixture that went stale) shows up here.  Both sides run on the CPU
over the oracle kernels (oracle/pn2_ext_oracle: the reference's CUDA extension restated), same weights, same numpy stream."""
import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import _ref_shims  # noqa: E402

pytestmark = [pytest.mark.refpy,
              pytest.mark.skipif(not _ref_shims.reference_available(), reason="reference tree not present (GPU box)")]

SEED = 4242          # scenes 4242 / 4243: used by no fixture


@pytest.fixture(scope="module")
def reference():
    saved_cuda = torch.Tensor.cuda
    saved_modules = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("multi_model", "dataset_utils", "pn2_ext",
                                                                                  "dgcnn_ext", "open3d")}
    mods = _ref_shims.import_reference()
    yield mods
    # undo the shims: the repo's own import-path aliases must serve every other test of the session
    torch.Tensor.cuda = saved_cuda
    for name in [m for m in sys.modules if m.split(".")[0] in ("multi_model", "dataset_utils", "pn2_ext", "dgcnn_ext", "open3d")]:
        del sys.modules[name]
    sys.modules.update(saved_modules)


def test_mirror_equals_the_live_reference_graph(reference, oracle_backend):
    sn, grn, grd = reference
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    from regnet_for_3d_grasping_amd.get_regiondataset import get_grasp_allobj
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
    B, N = 2, 6144
    pc = synthetic.make_batch(SEED, B, N)
    params, gparams = pipeline.PARAMS, pipeline.GRIPPER_PARAMS

    def build(score_cls, region_cls):
        s = score_cls(training=True)
        s.load_state_dict(synthetic.seeded_state_dict(s, 5))
        r = region_cls(training=True, group_num=params[2], gripper_num=64, grasp_score_threshold=0.5, radius=gparams[2],
                       reg_channel=10)
        r.load_state_dict(synthetic.seeded_state_dict(r, 6))
        return s.eval(), r.eval()

    ref_s, ref_r = build(sn.ScoreNetwork, grn.GripperRegionNetwork)
    mir_s, mir_r = build(ScoreNetwork, GripperRegionNetwork)
    assert type(ref_s).__module__.startswith("multi_model.") and os.path.abspath(
        sys.modules[type(ref_s).__module__].__file__).startswith(_ref_shims.REFERENCE_ROOT)
    synthetic.calibrate_score_head(mir_s, pc)
    ref_s.load_state_dict(mir_s.state_dict())        # same calibrated bn_score on both sides
    outs = []
    for s_net, r_net, grasp_allobj in ((ref_s, ref_r, grd.get_grasp_allobj), (mir_s, mir_r, get_grasp_allobj)):
        np.random.seed(77)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            feat, score, _ = s_net(pc)
            g = grasp_allobj(pc, score, params, [])
            res = r_net(g[3], g[5], g[2], g[4], g[0], g[1], pc, feat, gparams, None, [])
        outs.append((feat, score, g, res, int(np.random.randint(0, 2 ** 31 - 1))))
    (f0, s0, g0, r0, n0), (f1, s1, g1, r1, n1) = outs
    assert float((s0 - s1).abs().max()) <= 1e-6 and float((f0 - f1).abs().max()) <= 1e-5 * max(1.0, float(f0.abs().max()))
    assert int((s0 > 0.5).sum()) > 128                    # a real centre selection
    for k in (1, 2, 4):                                   # centre ids, small groups, large groups: identical indices
        assert torch.equal(g0[k].long(), g1[k].long()), k
    assert n0 == n1                                       # numpy's stream consumed identically
    assert r0[0].shape == r1[0].shape and float((r0[0] - r1[0]).abs().max()) <= 1e-5      # decoded stage-2 grasps
    assert (r0[6] is None) == (r1[6] is None)
    if r0[6] is not None:
        assert r0[6].shape == r1[6].shape and float((r0[6] - r1[6]).abs().max()) <= 1e-5   # refine-stage class-1 grasps
