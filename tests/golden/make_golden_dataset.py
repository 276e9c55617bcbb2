"""Dataset fixtures from the REFERENCE's ScoreDataset (authoring container only):

    python tests/golden/make_golden_dataset.py      # writes tests/golden/s5_dataset.npz

Synthetic records (tests/golden_util.dataset_records: seeded, 90 files, 150..400 points each) are written to
a temporary tree, the reference's ``dataset_utils/scoredataset.py`` class is run on them for every split tag
of both directory layouts, and the file lists plus three items per split (both resampling modes) are stored.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_shims  # noqa: E402

sys.path.insert(0, _ref_shims.REPO_ROOT)
import golden_util  # noqa: E402


def main():
    _ref_shims.import_reference()
    import importlib
    saved = list(sys.path)
    sys.path[:] = [_ref_shims.REFERENCE_ROOT] + [e for e in sys.path if os.path.abspath(e or ".") != _ref_shims.REPO_ROOT]
    try:
        for name in [m for m in sys.modules if m.split(".")[0] == "dataset_utils"]:
            del sys.modules[name]
        ref = importlib.import_module("dataset_utils.scoredataset")
    finally:
        sys.path[:] = saved
    assert os.path.abspath(ref.__file__).startswith(_ref_shims.REFERENCE_ROOT), ref.__file__
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        roots = golden_util.dataset_records(tmp)
        for case, (root, tag, seed, n_points) in golden_util.DATASET_CASES.items():
            ds = ref.ScoreDataset(n_points, roots[root], tag, seed, [0.06, 0.08])
            out[case + "/names"] = np.array([str(n) for n in ds.data_name])
            np.random.seed(seed + 1)
            for i in golden_util.DATASET_ITEMS:
                view, score, label, path, width = ds[i % len(ds)]
                out["%s/item%d/view" % (case, i)] = view.astype(np.float32)
                out["%s/item%d/score" % (case, i)] = score.astype(np.float32)
                out["%s/item%d/label" % (case, i)] = label.astype(np.float32)
                out["%s/item%d/path" % (case, i)] = np.array(os.path.relpath(path, tmp))
            out[case + "/rng_after"] = np.random.get_state()[1][:8].astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "s5_dataset.npz"), **out)
    print("wrote s5_dataset.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "s5_dataset.npz")), "bytes")


if __name__ == "__main__":
    main()
