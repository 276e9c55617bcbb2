"""The fenced split-products experiment (csrc/tsplit.hip, ``conv1x1_train.SPLIT_PRODUCTS``, off by default): forward, forward on a
pending BatchNorm + ReLU and input gradient of a 1x1 convolution (pn2_utils/nn/modules/conv.py:20-36) with every operand as three
bf16 pieces and six products on the bf16 matrix pipe -- as close to a float64 evaluation as the exact-fp32 kernels are."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,Co,Ci,L", [(2, 128, 128, 4096), (1, 256, 272, 1000), (3, 528, 64, 260), (1, 1024, 1024, 512), (2, 16, 48, 36)])
def test_split_products_match_float64_as_well_as_fp32_does(B, Co, Ci, L):
    from regnet_for_3d_grasping_amd import conv1x1_train as c
    assert c.SPLIT_PRODUCTS is False                     # never a default path
    g = torch.Generator().manual_seed(Co + Ci + L)
    x = torch.relu(torch.randn(B, Ci, L, generator=g) + 0.3).to(DEV)
    dy = torch.randn(B, Co, L, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, generator=g) / Ci ** 0.5).to(DEV)
    scale = (torch.rand(Ci, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(Ci, generator=g) * 0.2).to(DEV)
    bn = torch.relu(scale.double()[None, :, None] * x.double() + shift.double()[None, :, None])
    want = {"fwd": torch.einsum("oc,bcl->bol", w.double(), x.double()), "fwd_bn": torch.einsum("oc,bcl->bol", w.double(), bn),
            "dgrad": torch.einsum("oc,bol->bcl", w.double(), dy.double())}
    c.SPLIT_PRODUCTS = True
    try:
        got = {"fwd": c.native_fwd(x, w), "dgrad": c.native_dgrad(w, dy)}
        if Ci <= 1024:
            got["fwd_bn"] = c._split(0, w, x, B, Co, Ci, L, scale, shift, 1)
    finally:
        c.SPLIT_PRODUCTS = False
    for op, out in got.items():
        err = float((out.double() - want[op]).abs().max())
        # what a plain fp32 evaluation (one rounding per product and per addition) is allowed: ~ sqrt(K) ulps of the sum's size
        K = Co if op == "dgrad" else Ci
        bound = 4e-7 * K ** 0.5 * float(want[op].abs().max()) + 1e-6
        assert err <= bound, (op, err, bound)
        if c._native_ok(B, Co, Ci, L):               # ... and as close as the exact-fp32 kernel of csrc/tgemm.hip on the same operands
            exact = c.native_fwd(x, w) if op == "fwd" else c.native_dgrad(w, dy) if op == "dgrad" else c.native_fwd_bnrelu(x, w, scale, shift, 1) if Ci <= 512 and L % 16 == 0 else None
            if exact is not None:
                err32 = float((exact.double() - want[op]).abs().max())
                assert err <= 2.0 * err32 + 1e-6, (op, err, err32)


def test_split_level1_block_matches_the_exact_kernel_and_float64():
    """csrc/sa_split.hip (``fused.SPLIT_PRODUCTS``, off by default): the level-1 set-abstraction block (pointnet2.py:40-42) with
    all three layers on the bf16 matrix pipe, against the exact-fp32 ``sa_chain_kernel`` and against a float64 evaluation of the same
    packed layers -- and, end to end on the S8 scenes, the scores against the float64 fixture: the experiment's bar is "no further
    from float64 than the exact path"."""
    import os

    import numpy as np

    from . import golden_util as gu
    from regnet_for_3d_grasping_amd import fused, synthetic
    assert fused.SPLIT_PRODUCTS is False
    m7 = gu.meta_full()
    with open(os.path.join(gu.GOLDEN, "s8_meta.json")) as f:
        import json
        cfg = json.load(f)["cfg"]
    truth = np.load(os.path.join(gu.GOLDEN, "s8_score_fp64.npz"))
    net = gu.build_scorenet_full(m7, DEV)
    pc = synthetic.make_batch(cfg["scene_seed"], cfg["B"], cfg["N"]).to(DEV)
    errs = {}
    feats = {}
    for flag in (False, True):
        fused.SPLIT_PRODUCTS = flag
        try:
            with torch.no_grad():
                feat, score, _ = net(pc)
        finally:
            fused.SPLIT_PRODUCTS = False
        errs[flag] = float(np.abs(score.cpu().numpy().astype(np.float64) - truth["score"]).max())
        feats[flag] = feat
    print("S8 scores vs float64: exact path %.3e, level-1 block on split products %.3e" % (errs[False], errs[True]))
    assert errs[True] <= 1e-4
    assert errs[True] <= errs[False] * 1.10 + 2e-6          # no further from float64 than the exact path (10 % + 2e-6 of slack: both are
    #                                                       # fp32 evaluations whose maxima sit on different points)
    assert not torch.equal(feats[True], feats[False])      # (the switch really took the other kernel)
    assert float((feats[True] - feats[False]).abs().max()) <= 1e-4
