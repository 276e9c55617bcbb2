"""Device numpy-stream draws vs the host implementation on the counts of a real grouping pass: first differing row."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import get_regiondataset as grd, np_random, region_ops, synthetic
DEV = "cuda:0"
N = 12000
pc = synthetic.make_batch(9000, 1, N).to(DEV)
rng = np.random.default_rng(4)
score = torch.from_numpy(rng.uniform(0, 1, (1, N)).astype(np.float32)).to(DEV)
params = [4000, 0.5, 256, 0.1, 2048, 0.8, 0.08, 0.01, 0.06]
centre, idx = grd._select_score_center(pc, score, 4000, 0.5)
for size, rt in ((256, 0.1), (2048, 0.8)):
    radius = grd.group_radius(0.08, 0.01, 0.06, rt)
    cand, counts = region_ops.radius_candidates(pc, centre, radius)
    c = counts.cpu().numpy()
    print("size", size, "rows", c.size, "counts min/mean/max", c.min(), c.mean(), c.max(), "rows n>=size", int((c >= size).sum()))
    np.random.seed(21)
    want, wv = np_random.choice_rows(c, size, 0)
    after = np.random.randint(0, 2 ** 31 - 1, 3)
    np.random.seed(21)
    got, gv = np_random.choice_rows_device(counts, size, 0, N)
    np_random.flush()
    got = got.cpu().numpy()
    after2 = np.random.randint(0, 2 ** 31 - 1, 3)
    bad = np.nonzero((got != want).reshape(-1, size).any(1))[0]
    print("  rows differing:", len(bad), "first", bad[:5], "state equal", np.array_equal(after, after2))
    if len(bad):
        r = bad[0]
        w, g = want.reshape(-1, size)[r], got.reshape(-1, size)[r]
        k = np.nonzero(w != g)[0]
        print("  row", r, "n", c.reshape(-1)[r], "first differing slot", k[:5], "want", w[k[:5]], "got", g[k[:5]])
