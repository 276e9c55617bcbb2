"""the dataset record format + loader against fixtures produced by the REFERENCE's
``dataset_utils/scoredataset.py`` (tests/golden/make_golden_dataset.py): file split incl. the CPython set order of
the held-out part, per-item resampling in both modes, colour noise, tanh, and numpy RNG consumption."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_util  # noqa: E402


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "s5_dataset.npz"))


@pytest.fixture(scope="module")
def roots(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("records"))
    return tmp, golden_util.dataset_records(tmp)


@pytest.mark.parametrize("case", sorted(golden_util.DATASET_CASES))
def test_dataset_matches_reference(case, golden, roots):
    from dataset_utils.scoredataset import ScoreDataset      # the reference's import path
    tmp, trees = roots
    root, tag, seed, n_points = golden_util.DATASET_CASES[case]
    ds = ScoreDataset(n_points, trees[root], tag, seed, [0.06, 0.08])
    assert [str(n) for n in ds.data_name] == [str(n) for n in golden[case + "/names"]]
    np.random.seed(seed + 1)
    for i in golden_util.DATASET_ITEMS:
        view, score, label, path, width = ds[i % len(ds)]
        assert view.shape == (n_points, 6) and view.dtype == np.float32
        assert np.array_equal(view, golden["%s/item%d/view" % (case, i)])
        assert np.array_equal(score, golden["%s/item%d/score" % (case, i)])
        assert np.array_equal(label, golden["%s/item%d/label" % (case, i)])
        assert os.path.relpath(path, tmp) == str(golden["%s/item%d/path" % (case, i)])
        assert width.dtype == np.float32 and width.tolist() == pytest.approx([0.06, 0.08])
    assert np.array_equal(np.random.get_state()[1][:8].astype(np.int64), golden[case + "/rng_after"])


def test_records_feed_the_label_matcher(roots):
    """A record written by write_record carries the grasp keys _get_center_grasp reads (get_regiondataset.py:66-71)."""
    tmp, trees = roots
    rec = np.load(os.path.join(trees["training"], "training_data", "scene_0000.p"), allow_pickle=True)
    assert set(rec) >= {"view_cloud", "view_cloud_color", "view_cloud_score", "view_cloud_label", "frame",
                        "antipodal_score"}
    assert rec["frame"].shape[1:] == (4, 4) and rec["frame"].shape[0] == rec["antipodal_score"].shape[0]
