#!/usr/bin/env python
"""Level-1 neighbourhood sizes of the REFERENCE's own clouds -- the target of ``synthetic.make_scene(density="real")``.

AUTHORING-CONTAINER ONLY (reads /root/reference/test_file/*_predict/*.p; the pickles do not travel): 25 600-point random
subsamples of each file's ``points`` (SURVEY.md 8d "real-density variant"), furthest point sampling of 5 120 centroids and
the r = 0.02, K = 64 ball query through the CPU oracle (oracle/pn2_ext_oracle: the reference kernels restated), and the
distribution of the member counts.  Writes tests/golden/real_density_hist.json (data: a 65-bin histogram per file and
their mean), which tests and DESIGN.md quote.

    python scripts/real_density_hist.py
"""
import glob
import json
import os
import pickle
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REFERENCE = os.environ.get("REGNET_REFERENCE_ROOT", "/root/reference")


def counts_of(xyz, centroids=5120, radius=0.02, K=64):
    from oracle import pn2_ext_oracle as ext
    pts = torch.from_numpy(xyz.astype(np.float32).T[None]).contiguous()          # (1, 3, N)
    ctr = ext.farthest_point_sample(pts, centroids)
    cx = torch.gather(pts, 2, ctr[:, None, :].expand(1, 3, centroids))
    _, cnt = ext.ball_query(pts, cx, radius, K)
    return cnt.reshape(-1).numpy()


def summary(c):
    return {"mean": round(float(c.mean()), 2), "le16": round(float((c <= 16).mean()), 4), "le32": round(float((c <= 32).mean()), 4),
            "le48": round(float((c <= 48).mean()), 4), "eq64": round(float((c == 64).mean()), 4)}


def main():
    files = sorted(glob.glob(os.path.join(REFERENCE, "test_file", "*_predict", "*.p")))
    assert files, "no reference clouds under %s" % REFERENCE
    out = {"what": "level-1 ball-query member counts (r = 0.02, K = 64, 5 120 FPS centroids) of 25 600-point subsamples of the "
                   "reference's test_file/*_predict/*.p['points']; CPU oracle; seeds 0..1 per file",
           "files": {}}
    hist_sum = np.zeros(65)
    for f in files:
        with open(f, "rb") as fh:
            data = pickle.load(fh, encoding="latin1")
        pts = np.asarray(data["points"] if "points" in data else data["view_cloud"])[:, :3]
        pts = pts[np.isfinite(pts).all(1)]
        per_file = np.zeros(65)
        for seed in range(2):
            rng = np.random.default_rng(seed)
            sel = rng.choice(len(pts), 25600, replace=len(pts) < 25600)
            c = counts_of(pts[sel])
            per_file += np.bincount(c, minlength=65)[:65]
        name = os.path.relpath(f, os.path.join(REFERENCE, "test_file"))
        cc = np.repeat(np.arange(65), per_file.astype(np.int64))
        out["files"][name] = dict(summary(cc), points_in_file=int(len(pts)),
                                  extent=[round(float(v), 3) for v in (pts.max(0) - pts.min(0))])
        print(name, out["files"][name])
        hist_sum += per_file / per_file.sum()
    hist = hist_sum / len(files)
    cc = np.repeat(np.arange(65), np.round(hist * 1e6).astype(np.int64))
    out["mean_of_files"] = summary(cc)
    out["histogram_mean_of_files"] = [round(float(h), 6) for h in hist]
    print("mean of files", out["mean_of_files"])
    with open(os.path.join(REPO, "tests", "golden", "real_density_hist.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")


if __name__ == "__main__":
    main()
