"""GPU parity tests proper: every op of the HIP path, called through the C ABI (via the
``pn2_ext`` binding), against the CPU oracle on the same seeded inputs.

Bar: bit-exact for every index / count tensor; floats within the tolerance written in each test.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ext():
    from regnet_for_3d_grasping_amd import pn2_ext
    return pn2_ext


@pytest.fixture(scope="module")
def orc():
    from oracle import pn2_ext_oracle
    return pn2_ext_oracle


def cloud(seed, B, N, dup=0.0, grid=None):
    """(B,3,N) float32.  dup: fraction of points that are exact duplicates of earlier ones;
    grid: snap coordinates to a lattice so equal distances (ties) are frequent."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.4, 0.4, (B, N, 3)).astype(np.float32)
    if grid:
        p = (np.round(p / grid) * grid).astype(np.float32)
    if dup > 0:
        for b in range(B):
            k = int(N * dup)
            if k and N > 1:
                dst = rng.choice(N, k, replace=False)
                src = rng.integers(0, N, k)
                p[b, dst] = p[b, src]
    return torch.from_numpy(p).transpose(1, 2)  # (B,3,N) view of an AoS buffer (stride 3)


def layouts(x):
    """The same (B,3,N) values as: AoS view (as built), SoA contiguous, and the stride-6 view
    ScoreNet passes (pc[:, :, :6].permute(0, 2, 1)[:, :3, :])."""
    B, _, N = x.shape
    six = torch.zeros(B, N, 6)
    six[:, :, :3] = x.transpose(1, 2)
    return [x, x.contiguous(), six.permute(0, 2, 1)[:, :3, :]]


FPS_CASES = [  # (B, N, M, dup, grid)
    (2, 1024, 256, 0, None), (2, 5120, 1024, 0, None), (1, 6144, 5120, 0, None), (2, 1, 1, 0, None),
    (3, 20, 7, 0, None), (1, 64, 64, 0, None), (2, 300, 300, 0.3, None), (2, 2048, 512, 0.5, None),
    (2, 4096, 1024, 0, 0.05), (1, 700, 200, 0, 0.1), (1, 17, 17, 0.5, 0.2), (1, 9000, 700, 0.2, 0.02),
    (1, 13000, 300, 0, None), (1, 20000, 300, 0.1, None),
    # spatially sorted kernel (N > 8192): heavy ties / duplicates / degenerate extents
    (1, 12000, 2000, 0.3, 0.05), (2, 25600, 1000, 0.5, 0.01), (1, 9000, 9000, 0.1, None), (1, 16000, 500, 0.9, 0.1),
]


@pytest.mark.parametrize("B,N,M,dup,grid", FPS_CASES)
def test_fps_bit_exact(ext, orc, B, N, M, dup, grid):
    x = cloud(100 + N + M, B, N, dup, grid)
    want = orc.farthest_point_sample(x, M)
    for v in layouts(x):
        got = ext.farthest_point_sample(v.to(DEV), M)
        assert got.dtype == torch.int64 and tuple(got.shape) == (B, M)
        assert torch.equal(got.cpu(), want), "FPS mismatch (N=%d M=%d)" % (N, M)


FPS_CLUSTER_CASES = [  # (B, N, M, dup, grid): long runs = fps_cluster_kernel (several exact picks per round, csrc/geometry.hip)
    (2, 25600, 1500, 0.5, 0.01),     # heavy duplicates + lattice ties: the exact one-pick path in between batched rounds
    (1, 10000, 1024, 0, 0.2),        # 125 distinct points, 1024 picks: all distances zero after 125 -> the reference repeats its pick
    (1, 12001, 1100, 0.1, None),     # N not a multiple of the 64-point cluster
    (1, 8193, 1024, 0, None), (1, 4097, 512, 0.2, 0.03), (2, 8192, 2048, 0, None),
    (1, 16000, 1200, 0, None), (1, 20000, 8192, 0.05, None),   # 16 / 20 slots per lane; M at the LDS pick buffer's capacity
    (1, 20480, 8193, 0, None),       # one more: the per-wave kernel (fps_sorted_kernel<20, 4>) takes over
    (3, 25600, 5120, 0, None),
]


@pytest.mark.parametrize("B,N,M,dup,grid", FPS_CLUSTER_CASES)
def test_fps_cluster_kernel_bit_exact(ext, orc, B, N, M, dup, grid):
    x = cloud(900 + N + M, B, N, dup, grid)
    want = orc.farthest_point_sample(x, M)
    got = ext.farthest_point_sample(x.to(DEV), M).cpu()
    assert torch.equal(got, want), "FPS mismatch (N=%d M=%d): first difference at %s" % (
        N, M, (got != want).nonzero()[:1].tolist())
    got2 = ext.farthest_point_sample(layouts(x)[2].to(DEV), M).cpu()       # the strided view ScoreNet passes
    assert torch.equal(got2, want)


def test_fps_cluster_kernel_flat_and_collinear_scenes(ext, orc):
    """Degenerate extents for the extent-driven Morton key (all 12 bits go to one or two axes) at long-run sizes."""
    rng = np.random.default_rng(19)
    p = rng.uniform(-1, 1, (1, 14000, 3)).astype(np.float32)
    p[:, :, 2] = 0.75
    x = torch.from_numpy(p).transpose(1, 2)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 1500).cpu(), orc.farthest_point_sample(x, 1500))
    p[:, :, 1] = -0.2
    x = torch.from_numpy(p).transpose(1, 2)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 1500).cpu(), orc.farthest_point_sample(x, 1500))
    x = torch.ones(1, 3, 9000) * 0.25                      # zero extent everywhere, every distance zero from the start
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 1024).cpu(), orc.farthest_point_sample(x, 1024))
    # many EQUAL maxima in different clusters at once (more than the 16 candidates the batched round can hold): a coarse lattice
    # whose points are all at the same distance from the first pick's neighbours, each lattice site repeated ~6 times
    g = torch.stack(torch.meshgrid(torch.arange(12.), torch.arange(12.), torch.arange(12.), indexing="ij"), -1).view(-1, 3) * 0.1
    x = g.repeat(6, 1)[torch.randperm(6 * 1728, generator=torch.Generator().manual_seed(5))].t().contiguous().view(1, 3, -1)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 1500).cpu(), orc.farthest_point_sample(x, 1500))


def test_fps_all_identical_points(ext, orc):
    x = torch.ones(2, 3, 130) * 0.25
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 40).cpu(), orc.farthest_point_sample(x, 40))
    x = torch.ones(1, 3, 10000) * 0.25   # sorted kernel, zero-extent bounding box
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 50).cpu(), orc.farthest_point_sample(x, 50))


def test_fps_sorted_kernel_planar_and_collinear(ext, orc):
    rng = np.random.default_rng(9)
    p = rng.uniform(-1, 1, (1, 11000, 3)).astype(np.float32)
    p[:, :, 2] = 0.75                      # all points in one z plane (flat table)
    x = torch.from_numpy(p).transpose(1, 2)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 800).cpu(), orc.farthest_point_sample(x, 800))
    p[:, :, 1] = -0.2                      # ... and on one line
    x = torch.from_numpy(p).transpose(1, 2)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 800).cpu(), orc.farthest_point_sample(x, 800))


def test_fps_full_size_25600(ext, orc):
    from regnet_for_3d_grasping_amd import synthetic
    pc = synthetic.make_batch(1000, 2, 25600)
    x = pc.permute(0, 2, 1)[:, :3, :]
    got = ext.farthest_point_sample(x.to(DEV), 5120).cpu()
    assert torch.equal(got, orc.farthest_point_sample(x, 5120))
    for b in range(2):  # property: FPS of distinct points never repeats an index
        assert len(set(got[b].tolist())) == 5120


@pytest.mark.parametrize("B,N,M", [(1, 30000, 200), (3, 51200, 700), (2, 60000, 300), (1, 102400, 150)])
def test_fps_multi_workgroup_path_above_resident_limit(ext, orc, B, N, M):
    """25 600 < N <= 102 400: 2..4 workgroups per scene exchanging their maxima through global memory."""
    x = cloud(7 + N, B, N)
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), M).cpu(), orc.farthest_point_sample(x, M))


@pytest.mark.parametrize("B,N,M,dup,grid", [(2, 51200, 5120, 0, None), (1, 30000, 1500, 0.2, None), (1, 76800, 2048, 0, None),
                                           (1, 102400, 1024, 0, None), (2, 40000, 1200, 0.4, 0.02), (1, 25601, 1024, 0, None),
                                           (1, 50000, 1300, 0, 0.25)])
def test_fps_cooperative_cluster_kernel_bit_exact(ext, orc, B, N, M, dup, grid):
    """Scenes beyond 25 600 points with long runs: 2..4 cooperating workgroups, several exact picks per exchange
    (fps_cluster_kernel<.., true>): slices of unequal length, duplicates and lattice ties across the slice boundaries (the
    exact one-pick path exchanges keys), more picks than distinct points (every distance zero: the reference repeats)."""
    x = cloud(1900 + N + M, B, N, dup, grid)
    want = orc.farthest_point_sample(x, M)
    got = ext.farthest_point_sample(x.to(DEV), M).cpu()
    assert torch.equal(got, want), "FPS mismatch (N=%d M=%d): first difference at %s" % (
        N, M, (got != want).nonzero()[:1].tolist())


CHAIN_CASES = [  # (B, N, (M1, M2, M3), dup, grid)
    (3, 25600, (5120, 1024, 256), 0.0, None),      # the network's three levels on generic clouds
    (2, 51200, (5120, 1024, 256), 0.0, None),      # level 1 on cooperating workgroups
    (2, 6144, (5120, 1024, 256), 0.0, None),       # the small golden configuration
    (2, 25600, (5120, 1024, 256), 0.3, None),      # duplicated points: zero distances and exact ties
    (2, 12000, (4096, 1024, 256), 0.0, 0.05),      # lattice: ties at the maximum are frequent
    (1, 30000, (2048, 1024, 300), 0.1, 0.02),      # cooperative + lattice + duplicates
    (2, 3000, (1024, 512, 64), 0.0, None),         # level 1 on a kernel that does not track ties ("unknown")
    (1, 9000, (8800, 8000, 3000), 0.0, None),      # nearly every point picked, long chains
]


@pytest.mark.parametrize("B,N,Ms,dup,grid", CHAIN_CASES)
def test_fps_chain_levels_equal_independent_sampling(ext, orc, B, N, Ms, dup, grid):
    """Levels 2 and 3 sample the level above's centroids in pick order; with the level above's ``first_tie`` handed down
    (pn2_ext.FpsChain) scenes without a tie get 0 .. M-1 without sampling.  Whatever the shortcut does, every level must
    equal the ORACLE's sampling of the same cloud (which knows nothing of the shortcut), bit for bit."""
    x = cloud(4200 + N + Ms[0], B, N, dup, grid)
    xyz_gpu, xyz_cpu = x.to(DEV), x.contiguous()
    prefix = None
    shortcuts = 0
    for level, M in enumerate(Ms):
        chain = ext.FpsChain(prefix)
        got = ext.farthest_point_sample(xyz_gpu, M, chain)
        want = orc.farthest_point_sample(xyz_cpu, M)
        assert torch.equal(got.cpu(), want), "level %d (N=%d M=%d): first difference at %s" % (
            level + 1, xyz_cpu.shape[2], M, (got.cpu() != want).nonzero()[:1].tolist())
        tie = chain.first_tie.cpu()
        assert tie.dtype == torch.int32 and tie.shape == (B,)
        if prefix is not None:
            took = prefix.cpu() >= M
            shortcuts += int(took.sum())
            for b in range(B):
                if took[b]:           # the shortcut's claim, checked against the oracle above; its certificate is handed down
                    assert torch.equal(want[b], torch.arange(M)) and int(tie[b]) == int(prefix[b])
        # a certificate must never claim more than the truth: wherever first_tie says "no tie among the first m picks", the
        # next level's oracle sampling of the picks must be 0 .. m-1 (checked by the next iteration through `want`)
        prefix = chain.first_tie
        xyz_gpu = torch.gather(xyz_gpu, 2, got[:, None, :].expand(B, 3, M))
        xyz_cpu = torch.gather(xyz_cpu, 2, want[:, None, :].expand(B, 3, M)).contiguous()
    print("chain sampling N=%d: %d of %d lower-level runs took the shortcut" % (N, shortcuts, 2 * B))
    if dup == 0.0 and grid is None and N >= 6144:
        # generic clouds: most scenes have no exact fp32 tie at a maximum among their first 1024 picks (about one in five
        # does: ~20 000 candidate distances per round in a 2^23-value binade), and nearly none among the first 256
        assert shortcuts >= B, "generic clouds: most lower-level runs should have skipped their sampling"


def test_fps_chain_is_what_the_network_uses(ext, orc, monkeypatch):
    """PointNet2Seg.sample_levels / plan / the no-plan forward hand the certificate down: three FPS calls per forward, the
    second and third with a prefix; all three index tensors equal the oracle's."""
    from regnet_for_3d_grasping_amd import pipeline, synthetic
    net, _ = pipeline.build_models(DEV)
    pc = synthetic.make_batch(7300, 2, 25600).to(DEV)
    calls = []
    orig = ext.farthest_point_sample

    def spy(points, m, chain=None):
        out = orig(points, m, chain)
        calls.append((points.detach().clone(), int(m), chain is not None and chain.prefix_ok is not None, out.clone()))
        return out
    monkeypatch.setattr(ext, "farthest_point_sample", spy)
    for run in ("sample_levels", "plan", "forward"):
        calls.clear()
        with torch.no_grad():
            if run == "sample_levels":
                net.sample_levels(pc)
            elif run == "plan":
                net.plan(pc)
            else:
                net(pc)
        assert [c[1] for c in calls] == [5120, 1024, 256], run
        assert [c[2] for c in calls] == [False, True, True], run
        for points, m, _, out in calls:
            assert torch.equal(out.cpu(), orc.farthest_point_sample(points.cpu().contiguous(), m)), (run, m)


def test_fps_cooperative_status_word_is_accumulated_and_raised_lazily(ext):
    """A cooperative launch's status word (bit 0: a poll ran out of budget, the workgroup stopped sampling) is OR-ed into a
    per-device flag on the launch's stream; ``raise_if_fps_failed`` reads it where the caller synchronises.  A healthy
    launch leaves it clear; a set flag raises once and is cleared."""
    x = cloud(77, 2, 51200).to(DEV)
    ext.farthest_point_sample(x, 1024)
    ext.raise_if_fps_failed()                       # nothing lost
    flag = ext._fps_flag(x.device)
    assert int(flag.item()) == 0
    flag.fill_(1)                                   # what a launch with a lost partner leaves behind
    with pytest.raises(RuntimeError, match="lost its partner"):
        ext.raise_if_fps_failed()
    ext.raise_if_fps_failed()                       # cleared by the raise


def test_fps_multi_workgroup_ties_and_duplicates(ext, orc):
    """Lattice points (many exactly equal distances) and duplicated points across the workgroup boundary."""
    g = torch.stack(torch.meshgrid(torch.arange(40.), torch.arange(40.), torch.arange(20.), indexing="ij"), -1).view(1, -1, 3)
    x = (g[:, torch.randperm(g.shape[1], generator=torch.Generator().manual_seed(3))] * 0.05).permute(0, 2, 1).contiguous()
    x = torch.cat([x, x[:, :, :4000]], dim=2)          # 36 000 points, 4 000 of them duplicates
    assert torch.equal(ext.farthest_point_sample(x.to(DEV), 400).cpu(), orc.farthest_point_sample(x, 400))


def test_fps_streaming_fallback(ext, orc):
    """More than 32 767 samples of a large scene: the round tag of the multi-workgroup exchange would wrap, so the
    streaming kernel (running distances in the workspace) takes over."""
    x = cloud(11, 1, 33500)
    got = ext.farthest_point_sample(x.to(DEV), 33000).cpu()
    assert torch.equal(got, orc.farthest_point_sample(x, 33000))


BQ_CASES = [  # (B, N1, N2, radius, K)
    (2, 6144, 1024, 0.05, 64), (2, 1024, 256, 0.32, 64), (1, 50, 10, 0.2, 5), (2, 300, 70, 0.15, 3),
    (1, 1000, 33, 0.9, 128), (1, 64, 64, 0.1, 1), (2, 2500, 500, 0.001, 16), (1, 10, 3, 10.0, 20),
]


@pytest.mark.parametrize("B,N1,N2,radius,K", BQ_CASES)
def test_ball_query_bit_exact(ext, orc, B, N1, N2, radius, K):
    x = cloud(N1 + N2, B, N1, dup=0.1)
    rng = np.random.default_rng(K)
    # half of the centroids are points of the cloud (as in SA), half are arbitrary (may be empty balls)
    c = x[:, :, torch.from_numpy(rng.integers(0, N1, N2))].clone()
    c[:, :, N2 // 2:] += 0.013
    wi, wc = orc.ball_query(x, c, radius, K)
    for v in layouts(x):
        gi, gc = ext.ball_query(v.to(DEV), c.to(DEV), radius, K)
        assert gi.dtype == torch.int64 and gc.dtype == torch.int64
        assert torch.equal(gc.cpu(), wc)
        assert torch.equal(gi.cpu(), wi)


def test_ball_query_radius_boundary_is_strict(ext, orc):
    # points exactly at distance r: d2 == r*r must be excluded (strict <, ball_query_kernel.cu:61)
    x = torch.tensor([[[0.0, 0.5, 0.25, 0.5000001]], [[0.0] * 4], [[0.0] * 4]]).view(1, 3, 4)
    c = torch.zeros(1, 3, 1)
    gi, gc = ext.ball_query(x.to(DEV), c.to(DEV), 0.5, 4)
    wi, wc = orc.ball_query(x, c, 0.5, 4)
    assert torch.equal(gi.cpu(), wi) and torch.equal(gc.cpu(), wc)
    assert wc.item() == 2 and wi.view(-1).tolist() == [0, 2, 0, 0]


def test_ball_query_full_size_level1(ext, orc):
    from regnet_for_3d_grasping_amd import synthetic
    pc = synthetic.make_batch(1003, 1, 25600)
    x = pc.permute(0, 2, 1)[:, :3, :]
    idx = orc.farthest_point_sample(x, 5120)
    c = torch.gather(x, 2, idx[:, None, :].expand(1, 3, 5120))
    gi, gc = ext.ball_query(x.to(DEV), c.to(DEV), 0.02, 64)
    wi, wc = orc.ball_query(x, c, 0.02, 64)
    assert torch.equal(gi.cpu(), wi) and torch.equal(gc.cpu(), wc)
    # properties: members are within the radius, the first `count` slots ascend, the rest repeat slot 0
    g = torch.gather(x.contiguous(), 2, gi.cpu().view(1, 1, -1).expand(1, 3, -1)).view(1, 3, 5120, 64)
    d2 = ((g - c.unsqueeze(-1)) ** 2).sum(1)
    assert bool((d2 < 0.02 * 0.02 + 1e-9).all())
    assert int(gc.min()) >= 1  # every centroid is a point of the cloud


NN_CASES = [(2, 1024, 256), (2, 5120, 1024), (1, 6144, 5120), (1, 10, 3), (2, 77, 5), (1, 300, 2100)]


@pytest.mark.parametrize("B,N1,N2", NN_CASES)
def test_three_nn_bit_exact(ext, orc, B, N1, N2):
    q = cloud(N1, B, N1)
    k = cloud(N2 + 1, B, N2, dup=0.3, grid=0.05)  # duplicates + lattice => exact distance ties
    wi, wd = orc.point_search(q, k, 3)
    for v in layouts(q):
        gi, gd = ext.point_search(v.to(DEV), k.to(DEV), 3)
        assert torch.equal(gi.cpu(), wi)
        assert torch.equal(gd.cpu(), wd)  # same fp32 op order => identical bits


def test_three_nn_query_equals_key(ext, orc):
    k = cloud(5, 1, 200)
    gi, gd = ext.point_search(k.to(DEV), k.to(DEV), 3)
    assert torch.equal(gi[0, :, 0].cpu(), torch.arange(200)) and float(gd[0, :, 0].abs().max()) == 0.0


def test_group_points_forward_backward(ext, orc):
    rng = np.random.default_rng(0)
    B, C, N1, N2, K = 2, 37, 500, 60, 16
    x = torch.from_numpy(rng.normal(size=(B, N1, C)).astype(np.float32)).transpose(1, 2)  # non-contiguous (B,C,N1)
    idx = torch.from_numpy(rng.integers(0, N1, (B, N2, K)))
    got = ext.group_points_forward(x.to(DEV), idx.to(DEV)).cpu()
    assert torch.equal(got, orc.group_points_forward(x, idx))  # pure gather: exact
    g = torch.from_numpy(rng.normal(size=(B, C, N2, K)).astype(np.float32))
    gb = ext.group_points_backward(g.to(DEV), idx.to(DEV), N1).cpu()
    # atomics: summation order differs from the oracle's; tolerance 1e-5 on O(K) sums of unit normals
    torch.testing.assert_close(gb, orc.group_points_backward(g, idx, N1), rtol=0, atol=1e-5)


def test_group_points_autograd_matches_torch_gather(ext):
    from regnet_for_3d_grasping_amd.pn2_utils import function as F_
    torch.manual_seed(1)
    B, C, N, K = 2, 4, 5, 3  # the reference's own self-check sizes (functions/gather_knn.py:26-55)
    feat = torch.rand(B, C, N, device=DEV)
    idx = torch.randint(0, N, (B, N, K), device=DEV)
    a = feat.clone().requires_grad_(True)
    b = feat.clone().requires_grad_(True)
    ya = torch.gather(a.unsqueeze(2).expand(B, C, N, N), 3, idx.unsqueeze(1).expand(B, C, N, K))
    yb = F_.group_points(b, idx)
    assert torch.allclose(ya, yb)
    ya.backward(torch.ones_like(ya))
    yb.backward(torch.ones_like(yb))
    assert torch.allclose(a.grad, b.grad)


def test_gather_knn_reference_selfcheck(ext):
    from regnet_for_3d_grasping_amd.pn2_utils.functions.gather_knn import gather_knn
    torch.manual_seed(1)
    B, C, N, K = 2, 4, 5, 3
    feat = torch.rand(B, C, N, device=DEV)
    idx = torch.randint(0, N, (B, N, K), device=DEV)
    a = feat.clone().requires_grad_(True)
    b = feat.clone().requires_grad_(True)
    ya = torch.gather(a.unsqueeze(2).expand(B, C, N, N), 3, idx.unsqueeze(1).expand(B, C, N, K))
    yb = gather_knn(b, idx)
    assert ya.allclose(yb)
    ya.backward(torch.ones_like(ya))
    yb.backward(torch.ones_like(yb))
    assert a.grad.allclose(b.grad)


def test_interpolate_forward_backward(ext, orc):
    rng = np.random.default_rng(3)
    B, C, M, N = 2, 70, 128, 900
    x = torch.from_numpy(rng.normal(size=(B, M, C)).astype(np.float32)).transpose(1, 2)
    idx = torch.from_numpy(rng.integers(0, M, (B, N, 3)))
    w = torch.from_numpy(rng.dirichlet(np.ones(3), (B, N)).astype(np.float32))
    got = ext.interpolate_forward(x.to(DEV), idx.to(DEV), w.to(DEV)).cpu()
    # the kernel may contract mul+add into fma; 3-term sums of O(1) values: 1e-6
    torch.testing.assert_close(got, orc.interpolate_forward(x, idx, w), rtol=0, atol=1e-6)
    g = torch.from_numpy(rng.normal(size=(B, C, N)).astype(np.float32))
    gb = ext.interpolate_backward(g.to(DEV), idx.to(DEV), w.to(DEV), M).cpu()
    torch.testing.assert_close(gb, orc.interpolate_backward(g, idx, w, M), rtol=0, atol=2e-5)


@pytest.mark.parametrize("B,C,N1,N2,K", [(2, 9, 5120, 1024, 64), (1, 130, 36864, 300, 8), (2, 5, 40000, 700, 16),
                                         (3, 1, 7, 5, 3)])
def test_group_points_backward_row_lengths(ext, orc, B, C, N1, N2, K):
    """K4 (grouping_kernel.cu:54-93) through both implementations: output rows that fit the LDS accumulators
    (N1 <= 36 864) and the global-atomic fallback beyond; gradient given as a non-contiguous view."""
    rng = np.random.default_rng(N1 + K)
    idx = torch.from_numpy(rng.integers(0, N1, (B, N2, K)))
    g = torch.from_numpy(rng.normal(size=(B, N2, K, C)).astype(np.float32)).permute(0, 3, 1, 2)   # (B,C,N2,K) view
    gb = ext.group_points_backward(g.to(DEV), idx.to(DEV), N1).cpu()
    torch.testing.assert_close(gb, orc.group_points_backward(g.contiguous(), idx, N1), rtol=0, atol=5e-5)


@pytest.mark.parametrize("B,C,M,N", [(2, 64, 5120, 25600), (1, 33, 36864, 5000), (2, 6, 40000, 9000), (2, 3, 3, 11)])
def test_interpolate_backward_row_lengths(ext, orc, B, C, M, N):
    """K7 (interpolate_kernel.cu:239-282): LDS-accumulator path (M <= 36 864) and the global-atomic fallback."""
    rng = np.random.default_rng(M + N)
    idx = torch.from_numpy(rng.integers(0, M, (B, N, 3)))
    w = torch.from_numpy(rng.dirichlet(np.ones(3), (B, N)).astype(np.float32))
    g = torch.from_numpy(rng.normal(size=(B, N, C)).astype(np.float32)).transpose(1, 2)           # (B,C,N) view
    gb = ext.interpolate_backward(g.to(DEV), idx.to(DEV), w.to(DEV), M).cpu()
    torch.testing.assert_close(gb, orc.interpolate_backward(g.contiguous(), idx, w, M), rtol=0, atol=5e-5)


def test_empty_and_error_cases(ext):
    x = torch.zeros(2, 3, 10, device=DEV)
    with pytest.raises(RuntimeError):
        ext.farthest_point_sample(x, 0)          # CHECK_GT(num_centroids, 0)
    with pytest.raises(RuntimeError):
        ext.farthest_point_sample(x, 11)         # CHECK_GE(num_points, num_centroids)
    with pytest.raises(RuntimeError):
        ext.farthest_point_sample(torch.zeros(2, 4, 10, device=DEV), 2)   # CHECK_EQ(size(1), 3)
    with pytest.raises(RuntimeError):
        ext.point_search(x, torch.zeros(2, 3, 2, device=DEV), 3)          # CHECK_GE(num_key, 3)
    with pytest.raises(RuntimeError):
        ext.point_search(x, x, 4)                # only 3-NN
    with pytest.raises(RuntimeError):
        ext.ball_query(x.cpu(), x, 0.1, 4)       # CHECK_CUDA
    idx, cnt = ext.ball_query(x, torch.zeros(2, 3, 0, device=DEV), 0.1, 4)  # no centroids
    assert tuple(idx.shape) == (2, 0, 4) and tuple(cnt.shape) == (2, 0)
    e = ext.farthest_point_sample(torch.zeros(0, 3, 10, device=DEV), 2)      # empty batch
    assert tuple(e.shape) == (0, 2)


def test_region_ops_match_oracle():
    from oracle import region_oracle
    from regnet_for_3d_grasping_amd import region_ops, synthetic
    from regnet_for_3d_grasping_amd.get_regiondataset import group_radius
    pc = synthetic.make_batch(1001, 2, 6144)
    rng = np.random.default_rng(5)
    centres = torch.gather(pc, 1, torch.from_numpy(rng.integers(0, 6144, (2, 64, 1))).expand(2, 64, 6))
    for r_time in (0.1, 0.8):
        R = group_radius(0.08, 0.01, 0.06, r_time)
        wc, wn = region_oracle.radius_candidates(pc, centres, R)
        gc, gn = region_ops.radius_candidates(pc.to(DEV), centres.to(DEV), R)
        assert torch.equal(gn.cpu(), wn)
        for b in range(2):
            for c in range(64):
                n = int(wn[b, c])
                assert torch.equal(gc[b, c, :n].cpu(), wc[b, c, :n])
        # resampled groups (get_regiondataset.py:331-352): positions into the candidate lists -> member ids + their points,
        # -1 everywhere for a centre without candidates; against the reference's tensor expressions
        G = 96
        pos = torch.stack([torch.stack([torch.from_numpy(rng.integers(0, max(int(wn[b, c]), 1), G)) for c in range(64)]) for b in range(2)])
        pos[1, 5] = -1
        pos[0, 63] = -1
        gi, gp = region_ops.resample_groups(pc.to(DEV), gc, pos.to(DEV))
        want_i = torch.gather(gc.cpu().long(), 2, pos.clamp(min=0))
        want_p = torch.gather(pc, 1, want_i.view(2, 64 * G, 1).expand(2, 64 * G, 6)).view(2, 64, G, 6)
        empty = pos < 0
        want_i[empty] = -1
        want_p[empty] = -1.0
        assert torch.equal(gi.cpu(), want_i) and torch.equal(gp.cpu(), want_p)
    # box crop
    n, G = 40, 1024
    pts = torch.from_numpy(rng.uniform(-0.07, 0.07, (n, G, 6)).astype(np.float32))
    q, _ = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    rot = torch.from_numpy(q.astype(np.float32))
    centre = torch.from_numpy(rng.uniform(-0.01, 0.01, (n, 3)).astype(np.float32))
    xl = torch.full((n,), 0.03)
    yl = torch.from_numpy(rng.uniform(0.02, 0.04, n).astype(np.float32))
    wc, wn = region_oracle.box_candidates(pts, centre, rot, xl, yl, 0.005)
    gc, gn = region_ops.box_candidates(pts.to(DEV), centre.to(DEV), rot.to(DEV), xl.to(DEV), yl.to(DEV), 0.005)
    assert torch.equal(gn.cpu(), wn) and int(wn.max()) > 0
    for i in range(n):
        assert torch.equal(gc[i, :int(wn[i])].cpu(), wc[i, :int(wn[i])])
    # gather + max
    feat = torch.from_numpy(rng.normal(size=(2 * 6144, 256)).astype(np.float32))
    rows = torch.from_numpy(rng.integers(0, 2 * 6144, (128, 256)))
    assert torch.equal(region_ops.gather_max(feat.to(DEV), rows.to(DEV)).cpu(), region_oracle.gather_max(feat, rows))


# ---- uniform-grid variants: must reproduce the exhaustive kernels bit for bit -------------------------
def _grid_vs_plain(monkeypatch, fn):
    from regnet_for_3d_grasping_amd import pn2_ext as ext
    monkeypatch.setattr(ext, "GRID_MIN_POINTS", 1 << 40)
    monkeypatch.setattr(ext, "GRID_MIN_POINTS_BALL", 1 << 40)
    plain = fn(ext)
    monkeypatch.setattr(ext, "GRID_MIN_POINTS", 1)
    monkeypatch.setattr(ext, "GRID_MIN_POINTS_BALL", 1)
    grid = fn(ext)
    return plain, grid


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["scene", "uniform", "line", "duplicates", "lattice", "far_queries", "big_coords"])
def test_grid_three_nn_matches_exhaustive(case, monkeypatch):
    import numpy as np
    from regnet_for_3d_grasping_amd import synthetic
    rng = np.random.default_rng(5)
    if case == "scene":
        pc = synthetic.make_batch(77, 2, 6000)[..., :3].numpy()
        q, k = pc, pc[:, ::5]
    elif case == "uniform":
        q = rng.uniform(-1, 1, (2, 3000, 3)); k = rng.uniform(-1, 1, (2, 700, 3))
    elif case == "line":          # degenerate extent on two axes
        k = np.zeros((1, 500, 3)); k[..., 0] = rng.uniform(0, 1, (1, 500)); q = rng.uniform(-0.2, 1.2, (1, 900, 3)) * [1, 0.01, 0.01]
    elif case == "duplicates":    # all keys identical -> ties resolved by index
        k = np.tile(np.array([[0.3, 0.2, 0.1]]), (1, 64, 1)).reshape(1, 64, 3); q = rng.uniform(0, 1, (1, 300, 3))
    elif case == "lattice":       # many exactly equal distances
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(1, -1, 3) * 0.25
        k = g[:, rng.permutation(g.shape[1])]; q = g[:, rng.permutation(g.shape[1])] + 0.125
    elif case == "far_queries":   # queries well outside the key bounding box
        k = rng.uniform(0, 0.1, (2, 400, 3)); q = rng.uniform(-5, 5, (2, 1000, 3))
    else:                         # coordinates in millimetres
        k = rng.uniform(0, 800, (1, 900, 3)) + 5000; q = rng.uniform(0, 800, (1, 2500, 3)) + 5000
    qt = torch.tensor(q, dtype=torch.float32, device="cuda").permute(0, 2, 1)
    kt = torch.tensor(k, dtype=torch.float32, device="cuda").permute(0, 2, 1)
    plain, grid = _grid_vs_plain(monkeypatch, lambda ext: ext.point_search(qt, kt, 3))
    assert torch.equal(plain[0], grid[0])
    assert torch.equal(plain[1], grid[1])


@pytest.mark.gpu
@pytest.mark.parametrize("case,radius,K", [("scene", 0.02, 64), ("scene", 0.08, 64), ("scene", 0.3, 64),
                                           ("scene", 0.05, 16), ("uniform", 0.15, 32), ("lattice", 0.25, 64),
                                           ("empty", 0.01, 8), ("big_coords", 40.0, 64), ("all", 10.0, 64)])
def test_grid_ball_query_matches_exhaustive(case, radius, K, monkeypatch):
    import numpy as np
    from regnet_for_3d_grasping_amd import synthetic
    rng = np.random.default_rng(6)
    if case == "scene":
        p = synthetic.make_batch(78, 2, 8000)[..., :3].numpy(); c = p[:, :700]
    elif case == "uniform":
        p = rng.uniform(-1, 1, (2, 5000, 3)); c = rng.uniform(-1.1, 1.1, (2, 333, 3))
    elif case == "lattice":       # points exactly at distance == radius must be excluded (strict <)
        g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(6), indexing="ij"), -1).reshape(1, -1, 3) * 0.25
        p = g[:, rng.permutation(g.shape[1])]; c = p[:, :200]
    elif case == "empty":
        p = rng.uniform(0, 1, (1, 3000, 3)); c = rng.uniform(3, 4, (1, 50, 3))
    elif case == "big_coords":
        p = rng.uniform(0, 800, (1, 4000, 3)) + 5000; c = p[:, :300]
    else:                         # every point inside every ball: first K indices
        p = rng.uniform(0, 1, (1, 2500, 3)); c = p[:, :40]
    pt = torch.tensor(p, dtype=torch.float32, device="cuda").permute(0, 2, 1)
    ct = torch.tensor(c, dtype=torch.float32, device="cuda").permute(0, 2, 1)
    plain, grid = _grid_vs_plain(monkeypatch, lambda ext: ext.ball_query(pt, ct, radius, K))
    assert torch.equal(plain[1], grid[1])
    assert torch.equal(plain[0], grid[0])


# ---- centre picker: compaction kernel + one padded sampling launch for all scenes -----------------------
@pytest.mark.gpu
def test_select_positive_matches_oracle():
    import numpy as np
    from oracle import region_oracle
    from regnet_for_3d_grasping_amd import region_ops
    rng = np.random.default_rng(9)
    pc = torch.tensor(rng.normal(size=(4, 5000, 6)), dtype=torch.float32)
    score = torch.tensor(rng.uniform(size=(4, 5000)), dtype=torch.float32)
    score[2] = 0.0            # no positive at all
    score[3, 7:] = 0.0        # a handful
    want = region_oracle.select_positive(pc, score, 0.5)
    got = region_ops.select_positive(pc.cuda(), score.cuda(), 0.5)
    assert torch.equal(got[2].cpu(), want[2])
    for b in range(4):
        n = int(want[2][b])
        assert torch.equal(got[0][b, :n].cpu(), want[0][b, :n])
        if n:
            assert torch.equal(got[1][b].cpu(), want[1][b])


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 10, 63, 64, 65, 511, 513, 1025, 25600])
def test_select_positive_segment_edges(N):
    """Clouds shorter than one wave's segment, one point past a segment boundary, and the bench size: every wave of the
    workgroup walks its own contiguous segment (csrc/region.hip: SP_WAVES)."""
    import numpy as np
    from oracle import region_oracle
    from regnet_for_3d_grasping_amd import region_ops
    rng = np.random.default_rng(N)
    pc = torch.tensor(rng.normal(size=(3, N, 6)), dtype=torch.float32)
    score = torch.tensor(rng.uniform(size=(3, N)), dtype=torch.float32)
    score[1] = 1.0             # every point positive
    score[2, :-1] = 0.0        # at most the last one
    want = region_oracle.select_positive(pc, score, 0.5)
    got = region_ops.select_positive(pc.cuda(), score.cuda(), 0.5)
    assert torch.equal(got[2].cpu(), want[2])
    for b in range(3):
        n = int(want[2][b])
        assert torch.equal(got[0][b, :n].cpu(), want[0][b, :n])
        if n:
            assert torch.equal(got[1][b].cpu(), want[1][b])


@pytest.mark.gpu
def test_centre_picker_batched_sampling_equals_per_scene(monkeypatch):
    import numpy as np
    from regnet_for_3d_grasping_amd import get_regiondataset as grd, synthetic
    pc = synthetic.make_batch(321, 4, 12000, device="cuda")
    rng = np.random.default_rng(4)
    score = torch.tensor(rng.uniform(size=(4, 12000)), dtype=torch.float32, device="cuda")
    score[1] *= 0.7            # fewer positives than the other scenes
    score[3, 40:] = 0.0        # <= 64 positives: the numpy branch
    out = []
    for pad_min in (1024, 1 << 40):
        monkeypatch.setattr(grd, "_FPS_PAD_MIN", pad_min)
        np.random.seed(12)
        out.append(grd._select_score_center(pc, score, 64, 0.5))
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][0], out[1][0])
    # duplicates of the start point (index 0 of the compacted list) are never selected
    assert int((out[0][1][:3] < 0).sum()) == 0


def test_fps_small_scene_kernels_deterministic_under_load():
    """Regression: in the register-resident FPS kernels the tie-break slot of the PREVIOUS round used to be re-armed
    before the round's first barrier, so a wave that ran ahead could wipe the winner under a slower wave (which then
    kept its old centroid).  Seen only in the short-scan instantiations (N <= 1024: two points per thread) when other
    kernels share the CUs -- e.g. level-3 sampling inside the pipeline, once in ~300 batches.  Many small scenes, a
    busy side stream, repeated: every run must equal the oracle."""
    from oracle import pn2_ext_oracle
    from regnet_for_3d_grasping_amd import pn2_ext
    rng = np.random.default_rng(11)
    B, N, M = 96, 1024, 256
    pts = torch.from_numpy(rng.uniform(-1, 1, (B, 3, N)).astype(np.float32))
    want = pn2_ext_oracle.farthest_point_sample(pts, M)
    x = pts.to(DEV)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    for rep in range(60):
        with torch.cuda.stream(side):
            for _ in range(2):
                a @ a
        got = pn2_ext.farthest_point_sample(x, M)
        assert torch.equal(got.cpu(), want), "repeat %d differs from the oracle" % rep
    torch.cuda.synchronize()


def test_gripper_frame_and_crop_pick_match_the_torch_expressions():
    """region_ops.gripper_frame (one launch) against gripper_region_network.gripper_frame (the reference's frame maths as ~35
    torch launches: gripper_region_network.py:447-506) incl. zero-axis fallbacks, and region_ops.crop_pick against the
    gather / where expressions of get_gripper_region_transform (:540-548)."""
    from regnet_for_3d_grasping_amd import region_ops
    from regnet_for_3d_grasping_amd.gripper_region_network import gripper_frame
    g = torch.Generator().manual_seed(21)
    grasp = torch.randn(700, 10, generator=g)
    grasp[:, 6] = (torch.rand(700, generator=g) - 0.5) * 6.5
    grasp[0, 3:6] = 0.0                                   # zero closing axis -> fallback [0, 1, 0]
    grasp[1, 3:6] = torch.tensor([0.0, 0.0, 1.0])         # axis_x degenerate -> fallback [1, 0, 0]
    grasp[2, 3:6] = torch.tensor([0.0, 0.0, -2.0])
    grasp = grasp.to(DEV)
    c0, r0 = gripper_frame(grasp)
    c1, r1 = region_ops.gripper_frame(grasp)
    assert torch.equal(c0, c1)
    torch.testing.assert_close(r1, r0, rtol=0.0, atol=5e-7)      # (torch's norm / bmm may contract; this kernel does not)
    assert torch.isfinite(r1).all()
    c2, r2 = region_ops.gripper_frame(grasp[:, :8][5:300])       # strided rows
    assert torch.equal(c2, c1[5:300]) and torch.equal(r2, r1[5:300])

    n, G, R = 300, 1024, 64
    count = torch.randint(0, G + 1, (n,), generator=g)
    cand = torch.full((n, G), 12345, dtype=torch.int32)          # slots beyond count: never-written garbage
    for i in range(n):
        cand[i, :count[i]] = torch.sort(torch.randperm(G, generator=g)[:count[i]])[0].int()
    valid = count > 5
    pos = torch.stack([torch.randint(0, max(int(count[i]), 1), (R,), generator=g) for i in range(n)])
    gi = torch.randint(0, 25600, (n, G), generator=g)
    cand, valid, pos, gi = cand.to(DEV), valid.to(DEV), pos.to(DEV), gi.to(DEV)
    index = torch.where(valid.view(n, 1), torch.gather(cand, 1, pos).long(), torch.zeros_like(pos))
    inall = torch.gather(gi, 1, index)
    minus1 = torch.full((1, 1), -1, dtype=torch.int64, device=DEV)
    index, inall = torch.where(valid.view(n, 1), index, minus1), torch.where(valid.view(n, 1), inall, minus1)
    i1, a1 = region_ops.crop_pick(cand, pos, valid, gi)
    assert torch.equal(i1, index) and torch.equal(a1, inall)


def test_stage2_and_refine_decode_kernels_match_the_tensor_expressions():
    """region_ops.stage2_decode / refine_decode (one launch each, inference) against the reference-shaped tensor code they
    replace -- GripperRegionNetwork.compute_loss / compute_loss_refine without labels, run here on CPU copies
    (gripper_region_network.py:69-90, :201-215): grasp tuples to 1e-6, arg-max picks / class flags / selections equal,
    ties in the class scores included (torch.max keeps the first maximum)."""
    from regnet_for_3d_grasping_amd import region_ops
    from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
    net = GripperRegionNetwork(training=True, group_num=256, gripper_num=64, grasp_score_threshold=0.5, radius=0.06,
                               reg_channel=10)
    g = torch.Generator().manual_seed(5)
    for n in (1, 64, 512, 333):
        x_cls = torch.randn(n, 4, generator=g)
        x_cls[::7, 1] = x_cls[::7, 3]                       # ties between two anchors
        x_cls[::11] = 0.25                                  # ... and among all four
        raw = torch.randn(n, 4, 10, generator=g)
        centres = torch.randn(n, 6, generator=g) * 0.3      # rows of (xyz | rgb): the kernel reads the first three
        sig = raw.clone()
        sig[:, :, 7:] = torch.sigmoid(sig[:, :, 7:])
        anchors = net._enumerate_anchors(centres[:, :3].float())
        want = net.compute_loss(sig, anchors, x_cls, None)[0]
        tmpl = net.templates.float().reshape(-1, 4).to(DEV)
        got_raw = region_ops.stage2_decode(x_cls.to(DEV), raw.to(DEV), centres.to(DEV), tmpl, net.radius, True)
        got_sig = region_ops.stage2_decode(x_cls.to(DEV), sig.to(DEV), centres.to(DEV), tmpl, net.radius, False)
        np.testing.assert_allclose(got_raw.cpu().numpy(), want.numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(got_sig.cpu().numpy(), want.numpy(), rtol=0, atol=2e-6)

        grasp = want.clone()
        r_cls = torch.randn(n, 2, generator=g)
        r_cls[::5, 1] = r_cls[::5, 0]                       # tie -> class 0
        r_reg = torch.randn(n, 10, generator=g) * 0.1
        ref = net.compute_loss_refine(grasp, r_cls, r_reg, None)        # CPU tensors: the tensor-code branch
        final, flags = region_ops.refine_decode(grasp.to(DEV), r_cls.to(DEV), r_reg.to(DEV), net.radius, net.grasp_score_thre)
        flags = flags.cpu().numpy().astype(bool)
        assert np.array_equal(np.nonzero(flags[0])[0], ref[3].numpy()) and np.array_equal(np.nonzero(flags[1])[0], ref[4].numpy())
        np.testing.assert_allclose(final.cpu().numpy()[flags[0]], ref[0].numpy(), rtol=0, atol=1e-6)
        with torch.no_grad():                                            # the network's own branch on GPU tensors
            got = net.compute_loss_refine(grasp.to(DEV), r_cls.to(DEV), r_reg.to(DEV), None)
        for a, b in zip(got[:5], ref[:5]):
            np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=0, atol=1e-6)
