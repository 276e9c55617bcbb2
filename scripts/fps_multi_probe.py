"""MEASURE (not argue) a 2-4-CU cooperative furthest point sampling at N = 25 600: the library's fps_multi_kernel -- the
kernel that serves scenes beyond one CU's register file -- forced onto single-CU-sized scenes by measurement builds
(-DFPS_FORCE_MULTI=G of the measurement twin csrc/build.py generates from csrc/geometry.hip + scripts/ablate/geometry_measure.patch, build_variant(measure=True)), against the default
single-workgroup fps_sorted_kernel<25>.  One subprocess per library; outputs are compared bit for bit.

    python scripts/fps_multi_probe.py build      # authoring container (hipcc cross-compiles)
    python scripts/fps_multi_probe.py            # GPU box
"""
import hashlib, json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LIBS = {"default (1 workgroup per scene)": None,
        "one barrier per round (-DFPS_ONE_BARRIER=1)": os.path.join(REPO, "scripts", "ablate", "libregnet_fps_1barrier.so"),
        "2 cooperating workgroups": os.path.join(REPO, "scripts", "ablate", "libregnet_fps2.so"),
        "4 cooperating workgroups": os.path.join(REPO, "scripts", "ablate", "libregnet_fps4.so")}

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from regnet_for_3d_grasping_amd.csrc import build
    for g, path in ((2, LIBS["2 cooperating workgroups"]), (4, LIBS["4 cooperating workgroups"])):
        build.build_variant(path, ["-DFPS_FORCE_MULTI=%d" % g], measure=True)
        print("built", path)
    build.build_variant(LIBS["one barrier per round (-DFPS_ONE_BARRIER=1)"], ["-DFPS_ONE_BARRIER=1"], measure=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from regnet_for_3d_grasping_amd import pn2_ext, synthetic
    out = {}
    for B in (1, 8):
        pc = synthetic.make_batch(1000, B, 25600).to("cuda:0")
        xyz = pc[:, :, :3].permute(0, 2, 1)
        for M in (5120, 64):
            for _ in range(2):
                idx = pn2_ext.farthest_point_sample(xyz, M)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                idx = pn2_ext.farthest_point_sample(xyz, M)
            e.record(); torch.cuda.synchronize()
            out["B%d M%d" % (B, M)] = {"ms": round(s.elapsed_time(e) / 5, 4),
                                       "sha": hashlib.sha256(idx.cpu().numpy().tobytes()).hexdigest()[:16]}
    print("RESULT " + json.dumps(out))
    sys.exit(0)

results = {}
for name, lib in LIBS.items():
    env = dict(os.environ)
    if lib:
        env["REGNET_HIP_LIB"] = lib
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    results[name] = json.loads(line[0][7:]) if line else {"error": r.stderr[-400:]}
base = results["default (1 workgroup per scene)"]
for name, res in results.items():
    for key, v in res.items():
        if isinstance(v, dict):
            rounds = int(key.split("M")[1]) - 1
            print("%-34s %-10s %8.3f ms  %.2f us/round  %s" % (name, key, v["ms"], v["ms"] * 1e3 / rounds,
                  "bit-identical" if v["sha"] == base[key]["sha"] else "DIFFERENT OUTPUT"))
        else:
            print(name, key, v)
