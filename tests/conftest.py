import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "refpy: cross-check against the reference's Python graph "
                                       "(authoring container only; skipped when /root/reference is absent)")


def pytest_collection_modifyitems(config, items):
    """``gpu``-marked tests need a real MI355X: on a host without one they are SKIPPED (not failed), so a plain
    ``pytest tests`` is green on CPU-only CI; the GPU box runs them with ``-m gpu``."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def oracle_backend(monkeypatch):
    """Swap the HIP-backed extension modules of the product package for the CPU oracle so the
    host-side mirror (modules / models / grouping) can be exercised without a GPU.  Test-only:
    the product never routes through the oracle."""
    from oracle import pn2_ext_oracle, region_oracle
    import regnet_for_3d_grasping_amd.get_regiondataset as grd
    import regnet_for_3d_grasping_amd.gripper_region_network as grn
    import regnet_for_3d_grasping_amd.pn2_utils.function as fn
    import regnet_for_3d_grasping_amd.pn2_utils.functions.gather_knn as gk

    monkeypatch.setattr(fn, "pn2_ext", pn2_ext_oracle)
    monkeypatch.setattr(gk, "dgcnn_ext", pn2_ext_oracle)
    monkeypatch.setattr(grd, "region_ops", region_oracle)
    monkeypatch.setattr(grn, "region_ops", region_oracle)
    return pn2_ext_oracle
