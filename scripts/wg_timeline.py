"""Per-workgroup timeline of sa_chain_kernel (library variant built with scripts/ablate/chain_trace.h, see csrc/sa_chain.hip), alone and while a
resident side kernel holds whole CUs the way a level-1 furthest-point-sampling launch does: how many workgroups are resident
on how many CUs, and where the idle CUs are.
build here:  python -c "from regnet_for_3d_grasping_amd.csrc import build; build.build_variant('scripts/ablate/libregnet_trace.so', build.CHAIN_TRACE)"
             (and scripts/ablate/libclock_probe.so, see clock_probe.hip)
run on GPU:  REGNET_HIP_LIB=scripts/ablate/libregnet_trace.so python scripts/wg_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from regnet_for_3d_grasping_amd import pipeline, synthetic, _lib

side = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libclock_probe.so"))
side.side_load_lds.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L = _lib.lib
L.regnet_debug_set_chain_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
score_net, _ = pipeline.build_models(dev)
pc = synthetic.make_batch(1000, 8, 25600, device=dev)
with torch.no_grad():
    plan = score_net.plan(pc)
    for _ in range(3):
        score_net(pc, plan=plan)
torch.cuda.synchronize()
NB = 8 * 5120 // 8
trace = torch.zeros((NB, 4), dtype=torch.int64, device=dev)
sink = torch.zeros(4, device=dev)
sst = torch.cuda.Stream(dev, priority=-1)


def run(label, blocks, mode=6):
    """mode 6: every side workgroup stays (the hardware deals them to the XCDs in rotation); 16 + x: only those dealt to XCD x."""
    torch.cuda.synchronize()
    with torch.no_grad():
        if blocks:   # first, on an idle chip: a workgroup that wants a whole CU is only placed when one drains completely
            side.side_load_lds(blocks, 1024, 100.0, mode, 150 * 1024, sink.data_ptr(), sst.cuda_stream)
        for _ in range(5):
            score_net(pc, plan=plan)             # clocks up: a single forward after an idle gap runs ~12 % slower
        trace.zero_()
        L.regnet_debug_set_chain_trace(trace.data_ptr())
        score_net(pc, plan=plan)
    torch.cuda.synchronize()
    L.regnet_debug_set_chain_trace(None)
    t = trace.cpu().numpy()
    t0, t1, hw, xcc = t[:, 0], t[:, 1], t[:, 2], t[:, 3] & 0xf
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    print("== %s: kernel span %.1f us, %d workgroups on %d distinct CUs, mean workgroup %.1f us" % (label, e.max(), NB, len(np.unique(cu)), (e - s).mean()))
    for x in range(8):
        m = xcc == x
        print("   XCC %d: %4d workgroups on %2d CUs, last end %7.1f" % (x, m.sum(), len(np.unique(cu[m])), e[m].max()))
    edges = np.linspace(0, e.max(), 11)
    occ = [float(((np.minimum(e, b) - np.maximum(s, a)).clip(min=0)).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])]
    print("   resident workgroups per tenth of the span:", " ".join("%.0f" % o for o in occ))
    ids, per_cu = np.unique(cu, return_counts=True)
    gaps = []
    for c in ids:
        m = np.nonzero(cu == c)[0]
        o = m[np.argsort(s[m])]
        gaps.extend((s[o][1:] - e[o][:-1]).tolist())
    gaps = np.asarray(gaps)
    # workgroups are handed out in class order (fused.chain3_order): duration by twentieth of the launch, and the trace's own
    # stamps say how long a workgroup takes from its first instruction (weight staging included) to its last
    d = e - s
    setup = (t[:, 3] >> 8) / 100.0
    print("   set-up (start -> weights staged + neighbourhood gathered, us) by twentieth:", " ".join("%.1f" % setup[i].mean() for i in np.array_split(np.arange(NB), 20)))
    parts = np.array_split(np.arange(NB), 20)
    print("   workgroup duration by twentieth of the block range (us):", " ".join("%.0f" % d[i].mean() for i in parts))
    print("   sum of workgroup durations / (CUs x span) = %.3f; sum of between-workgroup gaps / (CUs x span) = %.3f" % (
        d.sum() / (len(ids) * e.max()), gaps.sum() / (len(ids) * e.max())))
    print("   workgroups per CU: min %d max %d; idle time between consecutive workgroups of a CU: median %.2f us, p90 %.1f, max %.1f, sum per CU %.1f us"
          % (per_cu.min(), per_cu.max(), np.median(gaps), np.percentile(gaps, 90), gaps.max(), gaps.sum() / len(ids)))


run("alone", 0)
if os.environ.get("HELD", "1") == "1":
    run("1 CU held", 1)
    run("8 CUs held, one per XCD", 8)
    run("16 CUs held, two per XCD", 16)
    run("8 CUs of ONE XCD held", 64, 16 + 7)
