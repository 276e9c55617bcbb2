"""Import the reference's Python graph (read-only, /root/reference) over the CPU oracle.

AUTHORING-CONTAINER ONLY: /root/reference does not exist on the GPU box, so nothing in the
test-suite imports this module at run time -- it is used by make_golden.py (which generates the
committed fixtures) and by the optional ``-m refpy`` cross-checks that skip when the reference is
absent.  Three shims are needed:
  1. a module named ``pn2_ext`` / ``dgcnn_ext`` on sys.modules  -> the CPU oracle binding,
  2. a stub ``open3d`` module (dead import, get_regiondataset.py:10),
  3. ``torch.Tensor.cuda`` -> identity (gripper_region_network.py:40-41,66-67 call it unconditionally).
"""
import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("REGNET_REFERENCE_ROOT", "/root/reference")
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "multi_model"))


def import_reference():
    """Returns (score_network, gripper_region_network, get_regiondataset) reference modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from oracle import pn2_ext_oracle

    sys.modules["pn2_ext"] = pn2_ext_oracle
    sys.modules["dgcnn_ext"] = pn2_ext_oracle
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    torch.Tensor.cuda = lambda self, *a, **k: self
    # make sure `multi_model` / `dataset_utils` resolve to the REFERENCE, not to this repo
    for name in [m for m in sys.modules if m.split(".")[0] in ("multi_model", "dataset_utils")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        sn = importlib.import_module("multi_model.score_network")
        grn = importlib.import_module("multi_model.gripper_region_network")
        grd = importlib.import_module("dataset_utils.get_regiondataset")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return sn, grn, grd
