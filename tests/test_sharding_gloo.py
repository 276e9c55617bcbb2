"""N>1 path on CPU: two gloo ranks shard the scene stream, run the forward through the oracle
backend, and the union of their results equals a single-process run of the same scenes."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _scores_for(seeds, n_points):
    """A small set-abstraction + feature-propagation pass (the hot path's building blocks, sized so
    the three processes of this test finish in seconds) over the scenes with the given seeds."""
    from oracle.install import oracle_backend
    from regnet_for_3d_grasping_amd import synthetic
    from regnet_for_3d_grasping_amd.pn2_utils.modules import PointNetSAModule, PointnetFPModule
    torch.set_num_threads(2)
    torch.manual_seed(0)
    sa = PointNetSAModule(3, (16, 16, 32), 128, 0.1, 16, True).eval()
    fp = PointnetFPModule(32 + 3, (32, 16), 3).eval()
    pc = torch.from_numpy(np.stack([synthetic.make_scene(s, n_points) for s in seeds], 0))
    xyz, rgb = pc.permute(0, 2, 1)[:, :3, :], pc.permute(0, 2, 1)[:, 3:6, :]
    with oracle_backend(), torch.no_grad():
        new_xyz, feat = sa(xyz, rgb)
        return fp(xyz, new_xyz, rgb, feat)


def _worker(rank, world, port, n_points, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from regnet_for_3d_grasping_amd import sharding
    dist = sharding.init("gloo")
    r, _, w = sharding.env_world()
    assert (r, w) == (rank, world)
    seeds = sharding.scene_seeds(r, w, 1, first_seed=5000)
    score = _scores_for(seeds, n_points)
    dist.barrier()
    slowest = sharding.max_over_ranks(1.0 + rank)          # rank 1 pretends to be slower
    total = sharding.gather_counts(len(seeds))
    torch.save({"seeds": seeds, "score": score, "slowest": slowest, "total": total},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_ranks_shard_scenes_without_exchange(tmp_path):
    world, n_points = 2, 2048
    port = _free_port()
    # keep the two ranks (+ this process) from oversubscribing the host: OpenMP/MKL pools are sized at
    # import time in the spawned interpreters
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "2"
    mp.spawn(_worker, args=(world, port, n_points, str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, "rank%d.pt" % r)) for r in range(world)]
    assert parts[0]["seeds"] == [5000] and parts[1]["seeds"] == [5001]      # disjoint shards
    assert all(p["slowest"] == 2.0 for p in parts)                          # MAX over ranks
    assert all(p["total"] == 2 for p in parts)                              # whole-job scene count
    whole = _scores_for([5000, 5001], n_points)
    got = torch.cat([parts[0]["score"], parts[1]["score"]], 0)
    # per-scene results do not depend on which rank (or batch) a scene ran in (eval mode)
    torch.testing.assert_close(got, whole, rtol=0, atol=1e-6)
