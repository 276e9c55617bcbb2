#!/usr/bin/env python
"""Level-1 neighbourhood-size histogram of ``synthetic.make_scene(density="real")`` against the reference clouds' (CPU oracle).

    python scripts/real_density_fit.py [first_seed] [scenes]

Prints mean / <= 16 / <= 32 / <= 48 / == 64 of the generator next to tests/golden/real_density_hist.json's targets; used to fit
``synthetic.REAL_DENSITY_BANDS`` (edit, rerun)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "scripts"))


def main():
    from real_density_hist import counts_of, summary
    from regnet_for_3d_grasping_amd import synthetic
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    density = sys.argv[3] if len(sys.argv) > 3 else "real"
    cs = [counts_of(synthetic.make_scene(first + i, 25600, density)[:, :3]) for i in range(n)]
    c = np.concatenate(cs)
    with open(os.path.join(REPO, "tests", "golden", "real_density_hist.json")) as f:
        target = json.load(f)["mean_of_files"]
    print("generator (%s, seeds %d..%d):" % (density, first, first + n - 1), summary(c))
    print("reference clouds (mean of files):  ", target)
    hist = np.bincount(c, minlength=65) / len(c)
    print("33..40 %.3f  41..48 %.3f  49..63 %.3f" % (hist[33:41].sum(), hist[41:49].sum(), hist[49:64].sum()))


if __name__ == "__main__":
    main()
