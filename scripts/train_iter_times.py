import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from regnet_for_3d_grasping_amd import pipeline, synthetic
from regnet_for_3d_grasping_amd.gripper_region_network import GripperRegionNetwork
from regnet_for_3d_grasping_amd.score_network import ScoreNetwork
from regnet_for_3d_grasping_amd.train_step import RefineTrainer
dev="cuda:0"; B,N=8,25600
pc=synthetic.make_batch(1000,B,N)
records=[synthetic.make_grasp_labels(pc[b].numpy(),50+b) for b in range(B)]
target=torch.from_numpy(np.random.default_rng(2).uniform(0,1,(B,N)).astype(np.float32)).to(dev)
s=ScoreNetwork(training=True); s.load_state_dict(synthetic.seeded_state_dict(s,7))
r=GripperRegionNetwork(training=True,group_num=256,gripper_num=64,grasp_score_threshold=0.5,radius=0.06,reg_channel=10)
r.load_state_dict(synthetic.seeded_state_dict(r,11))
t=RefineTrainer(s.to(dev),r.to(dev),pipeline.PARAMS,pipeline.GRIPPER_PARAMS)
pc=pc.to(dev); np.random.seed(0)
import gc
if os.environ.get("NOGC")=="1": gc.disable()
if os.environ.get("NOGC")=="2": gc.freeze()
ahead=t.prefetch(pc); ts=[]
for i in range(24):
    torch.cuda.synchronize(); t0=time.perf_counter()
    nxt=t.prefetch(pc); loss,parts=t.step(pc,target,records,plan=ahead); ahead=nxt
    torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print(" ".join("%.1f"%x for x in ts))
