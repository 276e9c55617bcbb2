"""Checkpoint fixtures written by the REFERENCE's own classes (authoring container only):

    python tests/golden/make_golden_ckpt.py     # writes tests/golden/ckpt_score_7.model.gz, ckpt_region_7.model.gz

The reference saves whole objects (train.py:467-468).  Here its ScoreNetwork (wrapped in nn.DataParallel, as
utils.py:131 does for training) and GripperRegionNetwork are instantiated from /root/reference, every tensor of
their state is filled with the deterministic pattern ``pattern(key, shape)`` below (arithmetic, so the 28 MB
compress to a few hundred KB), saved with torch.save exactly as the reference does, and gzipped.
tests/test_checkpoint_cpu.py restores them WITHOUT the reference present and checks every tensor.
"""
import gzip
import io
import os
import sys
import zlib

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402


def pattern(key, shape, dtype):
    """Deterministic, cheap-to-compress fill: a short arithmetic cycle whose phase depends on the key."""
    n = 1
    for s in shape:
        n *= s
    phase = zlib.crc32(key.encode()) % 11
    base = ((torch.arange(n, dtype=torch.int64) + phase) % 13 - 6).to(torch.float32) / 16.0
    if dtype in (torch.int64, torch.int32):
        return torch.full(shape, phase, dtype=dtype)
    return base.view(shape).to(dtype) if n else torch.zeros(shape, dtype=dtype)


def fill(model):
    state = model.state_dict()
    for key, value in state.items():
        filled = pattern(key.replace("module.", ""), tuple(value.shape), value.dtype)
        if key.endswith("running_var"):
            filled = filled.abs() + 0.5
        value.copy_(filled)
    return model


def fill_circulant(model):
    """Epoch-8 fixtures: numpy-seeded O(1) weights (tests/golden_util.circulant_state) -- a well-conditioned network, so
    that the HIP path's scores can be held to 1e-4 on reference-pickled weights without a sensitivity allowance."""
    sys.path.insert(0, _ref_shims.REPO_ROOT)
    from tests.golden_util import circulant_state
    state = model.state_dict()
    for key, value in state.items():
        value.copy_(circulant_state(key.replace("module.", ""), tuple(value.shape), value.dtype))
    return model


def main():
    sn, grn, _ = _ref_shims.import_reference()
    torch.manual_seed(0)

    def fresh():
        return (torch.nn.DataParallel(sn.ScoreNetwork(training=True, k_obj=2)),
                grn.GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5,
                                         radius=0.06, reg_channel=10))
    score, region = (fill(m) for m in fresh())
    score8, region8 = (fill_circulant(m) for m in fresh())
    for model, name in ((score, "ckpt_score_7.model.gz"), (region, "ckpt_region_7.model.gz"),
                        (score8, "ckpt_score_8.model.gz"), (region8, "ckpt_region_8.model.gz")):
        if os.path.exists(os.path.join(HERE, name)) and "--force" not in sys.argv:
            print(name, "exists (kept; --force rewrites it)")
            continue
        cls = model.module.__class__ if hasattr(model, "module") else model.__class__
        assert cls.__module__.startswith("multi_model."), cls.__module__
        buf = io.BytesIO()
        torch.save(model, buf)                     # whole-object pickle, as train.py:467-468
        path = os.path.join(HERE, name)
        with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
            f.write(buf.getvalue())
        print(name, len(buf.getvalue()), "->", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
