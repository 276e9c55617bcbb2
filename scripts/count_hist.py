"""Distribution of the level-1 ball-query counts on the synthetic bench scenes (how many neighbourhood slots are padding)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from regnet_for_3d_grasping_amd import pn2_ext, synthetic
pc = synthetic.make_batch(1000, 8, 25600, device="cuda:0")
xyz = pc.permute(0, 2, 1)[:, :3, :]
ctr = pn2_ext.farthest_point_sample(xyz, 5120)
cx = torch.gather(xyz, 2, ctr[:, None, :].expand(8, 3, 5120))
_, cnt = pn2_ext.ball_query(xyz, cx, 0.02, 64)
c = cnt.float()
print("mean count %.1f  <=16: %.3f  <=32: %.3f  <=48: %.3f  ==64: %.3f" % (c.mean(), (c <= 16).float().mean(), (c <= 32).float().mean(), (c <= 48).float().mean(), (c == 64).float().mean()))
