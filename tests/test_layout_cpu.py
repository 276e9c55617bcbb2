"""Host-side layout logic of the round-2 kernels, checked without a GPU:
  * the LDS chunk swizzles of gemm2.h / tgemm.hip ([row][16 k] tiles) and rowchain.hip ([32][256] / [64][128] stages)
    put the 16 lanes of every ds_read_b128 service group on 16 distinct 16-byte slots of the 256-byte bank row
    (MI355X_MICROARCH.md, LDS table: groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32);
  * fused._swizzle_stage / _rowchain_pair_stream produce exactly the image the kernels' fragment offsets read
    (a numpy model of rc_frag_offsets + the stage order recovers every weight);
  * bench.py's kernel-name mapping follows csrc/mlp.hip:launch_gemm2's tile table."""
import numpy as np
import torch

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def _slots(byte_addresses):
    return sorted((a // 16) % 16 for a in byte_addresses)


def test_gemm2_tile_swizzle_is_bank_conflict_free():
    # [row][16 floats] dense rows (64 B); lane (fr = l & 31, fh = l >> 5) reads logical chunk 2 kk + fh of row base + fr,
    # stored at chunk c ^ ((row >> 2) & 3)
    for base in (0, 32, 64, 96, 128, 192):
        for kk in (0, 1):
            for group in GROUPS:
                addrs = []
                for l in group:
                    fr, fh = l & 31, l >> 5
                    row = base + fr
                    addrs.append(row * 64 + 16 * ((2 * kk + fh) ^ ((row >> 2) & 3)))
                assert _slots(addrs) == list(range(16)), (base, kk, group)


def _rc_offsets(row_floats):
    """rc_frag_offsets: per lane (j = l & 15, g = l >> 4) float offsets q[0..3]; step kt reads q[kt & 3] + 64 (kt >> 2)."""
    off = np.zeros((64, 4), dtype=np.int64)
    for l in range(64):
        j, g = l & 15, l >> 4
        for q in range(4):
            off[l, q] = j * row_floats + 4 * ((4 * q) ^ (g ^ j))
    return off


def test_rowchain_stage_swizzle_is_bank_conflict_free():
    for row_floats, steps in ((256, 16), (128, 8)):
        off = _rc_offsets(row_floats)
        for kt in range(steps):
            for tile in (0, 1):
                for group in GROUPS:
                    addrs = [4 * (tile * 16 * row_floats + off[l, kt & 3] + 64 * (kt >> 2)) for l in group]
                    assert _slots(addrs) == list(range(16)), (row_floats, kt, tile, group)


def test_swizzle_stage_matches_the_kernels_fragment_reads():
    from regnet_for_3d_grasping_amd import fused
    rng = np.random.default_rng(0)
    for rows, kc in ((32, 256), (64, 128)):
        W = torch.from_numpy(rng.normal(size=(rows, kc)).astype(np.float32))
        image = fused._swizzle_stage(W).numpy()
        off = _rc_offsets(kc)
        for tile in range(rows // 16):
            for kt in range(kc // 16):
                for l in range(64):
                    j, g = l & 15, l >> 4
                    a = tile * 16 * kc + off[l, kt & 3] + 64 * (kt >> 2)
                    got = image[a:a + 4]
                    want = W[tile * 16 + j, 16 * kt + 4 * g:16 * kt + 4 * g + 4].numpy()
                    assert np.array_equal(got, want), (rows, tile, kt, l)


def test_rowchain_pair_stream_order():
    """Per 128-channel group of the middle width: four A-stages (32 rows of W_A, all 256 columns) then N/64 B-stages
    (64 rows of W_B, that group's 128 columns)."""
    from regnet_for_3d_grasping_amd import fused

    def layer(n, k, seed):
        L = fused._Layer()
        L.N, L.K = n, k
        L.W = torch.from_numpy(np.random.default_rng(seed).normal(size=(n, k)).astype(np.float32))
        return L

    la, lb = layer(512, 256, 1), layer(256, 512, 2)
    stages = fused._rowchain_pair_stream(la, lb)
    assert len(stages) == 4 * (4 + 4)
    s = 0
    for ob in range(4):
        for u in range(4):
            assert torch.equal(stages[s], fused._swizzle_stage(la.W[128 * ob + 32 * u:128 * ob + 32 * u + 32, :]))
            s += 1
        for v in range(4):
            assert torch.equal(stages[s], fused._swizzle_stage(lb.W[64 * v:64 * v + 64, 128 * ob:128 * ob + 128]))
            s += 1
    assert all(st.numel() == 8192 for st in stages)


def test_bench_kernel_names_follow_the_tile_table():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    k = bench._mlp_layer_kernel
    assert k({"P": 512, "K": 1024, "N": 256}) == "mlp_gemm_kernel<0>"          # skinny: split-K heads
    assert k({"P": 8192, "K": 3, "N": 256}) == "mlp_gemm_kernel<0>"             # Kpad < 32
    assert k({"P": 131072, "K": 512, "N": 1024, "pool": 64}) == "gemm2_kernel<256,128,pool>"
    assert k({"P": 204800, "K": 256, "N": 256}) == "gemm2_kernel<256,128>"      # >= four rounds of the big tile
    assert k({"P": 40960, "K": 512, "N": 512}) == "gemm2_kernel<64,128>"        # >= 2 slabs of K: the slab-accumulating tile
    assert k({"P": 40960, "K": 128, "N": 512}) == "gemm2_kernel<128,128>"       # one slab, 640 big tiles: not enough rounds
    assert k({"P": 8192, "K": 1024, "N": 1024}) == "gemm2_kernel<64,128>"       # few tiles: the smallest one
    assert k({"P": 8192, "K": 1024, "N": 512}) == "gemm2_kernel<64,128>"
    assert k({"P": 40960, "K": 512, "N": 256}) == "gemm2_kernel<64,128>"
    assert k({"P": 2048, "K": 1024, "N": 1024}) == "gemm2_kernel<64,128>"
    assert k({"P": 204800, "K": 128, "N": 128}) == "gemm2_kernel<128,128>"
    assert k({"P": 204800, "K": 256, "N": 128}) == "gemm2_kernel<64,128>"


def test_density_matched_generator_reproduces_the_reference_histogram():
    """``synthetic.make_scene(density="real")``: the level-1 neighbourhood sizes (CPU oracle: FPS 5 120 + ball query r = 0.02,
    K = 64) of one scene against the reference clouds' histogram (tests/golden/real_density_hist.json, derived by
    scripts/real_density_hist.py in the authoring container); the default generator is untouched by the new keyword."""
    import hashlib
    import json
    import os
    import numpy as np
    import torch
    from oracle import pn2_ext_oracle as ext
    from regnet_for_3d_grasping_amd import synthetic
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden", "real_density_hist.json")) as f:
        doc = json.load(f)
    target = doc["mean_of_files"]
    assert abs(sum(doc["histogram_mean_of_files"]) - 1.0) < 1e-3 and len(doc["files"]) == 4
    scene = synthetic.make_scene(1001, 25600, density="real")
    assert scene.shape == (25600, 6) and scene.dtype == np.float32 and np.isfinite(scene).all()
    pts = torch.from_numpy(np.ascontiguousarray(scene[:, :3].T[None]))
    ctr = ext.farthest_point_sample(pts, 5120)
    cx = torch.gather(pts, 2, ctr[:, None, :].expand(1, 3, 5120))
    _, cnt = ext.ball_query(pts, cx, 0.02, 64)
    c = cnt.reshape(-1).float()
    assert abs(float(c.mean()) - target["mean"]) <= 2.5
    assert abs(float((c <= 32).float().mean()) - target["le32"]) <= 0.05
    assert abs(float((c <= 48).float().mean()) - target["le48"]) <= 0.05
    assert abs(float((c == 64).float().mean()) - target["eq64"]) <= 0.05
    # the fixtures' generator: same bytes as before the keyword existed
    assert hashlib.sha256(synthetic.make_scene(1000).tobytes()).hexdigest().startswith("8d35fe7b114468ee")
    assert np.array_equal(synthetic.make_batch(1000, 1, 512)[0].numpy(), synthetic.make_scene(1000, 512))


def test_experiment_patches_still_apply_to_the_product_sources():
    """scripts/ablate/*.patch are measured-and-not-kept kernels and measurement switches kept as patches against the product sources
    (DESIGN.md par. 13.3, 13.7): one that stops applying no longer describes the product.  ``historical_*`` are records of earlier
    rounds' experiments against sources that have moved on; ``geometry_measure.patch`` writes a twin file (csrc/build.py)."""
    import glob
    import os
    import shutil
    import subprocess
    import tempfile
    if shutil.which("patch") is None:
        import pytest
        pytest.skip("no patch(1) here")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    live = sorted(p for p in glob.glob(os.path.join(repo, "scripts", "ablate", "*.patch"))
                  if not os.path.basename(p).startswith("historical_") and os.path.basename(p) != "geometry_measure.patch")
    assert len(live) >= 4
    for path in live:
        r = subprocess.run(["patch", "--dry-run", "-s", "-p1", "-i", path], cwd=repo, capture_output=True, text=True)
        assert r.returncode == 0, (os.path.basename(path), r.stdout[-400:], r.stderr[-400:])
    from regnet_for_3d_grasping_amd.csrc import build
    with tempfile.TemporaryDirectory() as tmp:
        twin = build.measurement_twin("geometry.hip", tmp)
        assert os.path.getsize(twin) > os.path.getsize(os.path.join(repo, "regnet_for_3d_grasping_amd", "csrc", "geometry.hip"))
