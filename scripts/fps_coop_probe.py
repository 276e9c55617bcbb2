"""Level-1 sampling (5 120 of 25 600 points) on ONE workgroup per scene (the product) against 2 / 4 COOPERATING workgroups per scene
(fps_cluster_kernel<.., true>, the kernel that serves scenes beyond 25 600 points, forced onto 25 600 by measurement builds
-DFPS_COOP_MIN_N=20000 -DFPS_COOP_SLICE=12800 / 6400): milliseconds per launch and a digest of the picks.
    python scripts/fps_coop_probe.py build     # authoring container
    python scripts/fps_coop_probe.py           # GPU box"""
import hashlib, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIBS = {"1 workgroup per scene (product)": None,
        "2 cooperating workgroups": os.path.join(REPO, "gpurun_variant_fpscoop2.so"),
        "4 cooperating workgroups": os.path.join(REPO, "gpurun_variant_fpscoop4.so")}
if sys.argv[1:] == ["build"]:
    from regnet_for_3d_grasping_amd.csrc import build
    build.build_variant(LIBS["2 cooperating workgroups"], ["-DFPS_COOP_MIN_N=20000", "-DFPS_COOP_SLICE=12800"])
    build.build_variant(LIBS["4 cooperating workgroups"], ["-DFPS_COOP_MIN_N=20000", "-DFPS_COOP_SLICE=6400"])
    sys.exit(0)
if sys.argv[1:] == ["child"]:
    import torch
    from regnet_for_3d_grasping_amd import pn2_ext, synthetic
    for B in (1, 8, 64):
        pc = synthetic.make_batch(1000, B, 25600).to("cuda:0")
        xyz = pc.permute(0, 2, 1)[:, :3, :]
        for _ in range(3):
            idx = pn2_ext.farthest_point_sample(xyz, 5120)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            idx = pn2_ext.farthest_point_sample(xyz, 5120)
        e.record(); torch.cuda.synchronize()
        pn2_ext.raise_if_fps_failed()
        print("   B%-3d %.3f ms per launch, %.3f us per pick, picks %s" % (B, s.elapsed_time(e) / 10, s.elapsed_time(e) / 10 / 5.12,
                                                                          hashlib.sha256(idx.cpu().numpy().tobytes()).hexdigest()[:12]))
    sys.exit(0)
for name, lib in LIBS.items():
    env = dict(os.environ)
    if lib:
        env["REGNET_HIP_LIB"] = lib
    print(name)
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=300)
