"""Pipeline race hunt: on a differing batch, compare its geometry plan (every index tensor) with the sequential one."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regnet_for_3d_grasping_amd import pipeline, synthetic, fused
DEV = 'cuda:0'
fused.ROWCHAIN = os.environ.get("ROWCHAIN", "1") == "1"
net, region_net = pipeline.build_models(DEV)
seg = net.extrat_featurePN2
pc = synthetic.make_batch(2000, 4, 25600).to(DEV)
synthetic.calibrate_score_head(net, pc[:1])
order = [[0, 1, 2, 3], [3, 2, 1, 0], [1, 0, 3, 2]]
batches = [pc[o].contiguous() for o in order]
refs = []
with torch.no_grad():
    for b in batches:
        plan = seg.plan(b[:, :, :6].permute(0, 2, 1))
        f, s, _ = net(b, plan=plan)
        refs.append((plan, f, s))
torch.cuda.synchronize()

def flat(plan):
    out = {}
    for kind in ("sa", "fp"):
        for li, level in enumerate(plan[kind]):
            for k, v in level.items():
                if torch.is_tensor(v):
                    out["%s%d.%s" % (kind, li, k)] = v
    return out

bad = 0
for trial in range(int(os.environ.get("TRIALS", "200"))):
    # GRAPHS=1: the hipGraph replays of the geometry + feature stages (outputs only); default: launch by launch, plans kept
    pipe = pipeline.ForwardPipeline(net, region_net, with_region=False, graphs=os.environ.get("GRAPHS", "0") == "1")
    stash = []
    orig_features = pipe._features
    def feat(item, _o=orig_features):
        stash.append({k: v for k, v in flat(item["plan"]).items()} if "plan" in item else {})   # references keep the tensors alive
        return _o(item)
    pipe._features = feat
    outs = list(pipe.run(iter(batches)))
    torch.cuda.synchronize()
    for bi, out in enumerate(outs):
        plan0, f0, s0 = refs[bi]
        if not (torch.equal(out["all_feature"], f0) and torch.equal(out["score"], s0)):
            bad += 1
            print("trial", trial, "batch", bi, "feature/score differ")
            ref_flat = flat(plan0)
            for name, t in stash[bi].items():
                r = ref_flat[name]
                if t.shape != r.shape or not torch.equal(t, r):
                    d = (t != r)
                    per_scene = d.reshape(d.shape[0], -1).sum(1).tolist()
                    print("   plan tensor", name, tuple(t.shape), "differs; mismatches per scene", per_scene)
print("differing batches:", bad)
