#!/usr/bin/env python
"""rocprofv3 PMC passes (scripts/collect_mfma_pmc.sh) -> per-kernel-family counter averages and derived ratios."""
import collections
import csv
import glob
import json
import sys

FAMILY = [("gemm2_kernel<256, 128, 4, 2, 3, 2, true>", "gemm2_kernel<256,128,pool>"), ("gemm2_kernel<256, 128, 4, 2, 3, 2, false>", "gemm2_kernel<256,128>"),
          ("gemm2_kernel<128, 128", "gemm2_kernel<128,128>"), ("sa3_premul_chain_kernel", "sa3_premul_chain_kernel"), ("sa_premul_chain_kernel", "sa_premul_chain_kernel"), ("gemm2_kernel<64, 128", "gemm2_kernel<64,128>"), ("mlp_gemm_kernel<3", "mlp_gemm_kernel<3>"), ("mlp_gemm_kernel<0", "mlp_gemm_kernel<0>"), ("mlp_gemm_kernel", "mlp_gemm_kernel"), ("fp_head_chain_kernel", "fp_head_chain_kernel"),
          ("sa_chain_kernel", "sa_chain_kernel"), ("fps_cluster_kernel", "fps_cluster_kernel"), ("fps_sorted_kernel", "fps_sorted_kernel"),
          ("fps_resident_kernel", "fps_resident_kernel"), ("interp_affine_kernel", "interp_affine_kernel"),
          ("ball_query_grid_kernel", "ball_query_grid_kernel"), ("three_nn_grid_kernel", "three_nn_grid_kernel"),
          ("radius_group_kernel", "radius_group_kernel"), ("gather_max", "gather_max_kernel"), ("probe<", "mfma_peak_probe")]


def family(name):
    for pat, fam in FAMILY:
        if pat in name:
            return fam
    return None


def collect(directory):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam == "fp_head_chain_kernel" and r.get("Grid_Size") and int(r["Grid_Size"]) < 256 * 512:
                # the partial last round of row blocks (64 of 1 600) is its own launch on a side stream (fused.TAIL_SINK): a
                # quarter of the CUs for one block's time -- counted with the main launch it reads as matrix-pipe idleness
                fam = "fp_head_chain_kernel (tail launch: %d workgroups)" % (int(r["Grid_Size"]) // 512)
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {fam: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())}
            for fam, cs in acc.items()}


def main(cal_dir, g1_dir, g2_dir):
    cal = collect(cal_dir).get("mfma_peak_probe", {})
    cal_ratio = cal.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cal.get("GRBM_GUI_ACTIVE", 1), 1)
    g1, g2 = collect(g1_dir), collect(g2_dir)
    out = {"calibration": {"kernel": "scripts/ablate/mfma_peak (bare v_mfma_f32_32x32x2_f32 loops, all modes averaged)",
                           "mfma_busy_per_gui_active": round(cal_ratio, 3), "counters": {k: round(v) for k, v in cal.items()}},
           "families": {}}
    for fam in sorted(set(g1) | set(g2)):
        a, b = g1.get(fam, {}), g2.get(fam, {})
        row = {k: round(v) for k, v in a.items()}
        row.update({k: round(v) for k, v in b.items() if k not in row})
        gui = a.get("GRBM_GUI_ACTIVE", 0)
        if gui and cal_ratio:
            row["mfma_busy_frac_of_bare_loop"] = round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / gui / cal_ratio, 4)
        wc = a.get("SQ_WAVE_CYCLES", 0)
        if wc:
            row["wave_cycles_waiting_for_issue_frac"] = round(a.get("SQ_WAIT_INST_ANY", 0) / wc, 4)
            row["wave_cycles_parked_frac"] = round(a.get("SQ_WAIT_ANY", 0) / wc, 4)
        if b.get("SQ_INSTS_VALU") and b.get("GRBM_GUI_ACTIVE"):
            row["valu_insts_per_gui_cycle"] = round(b["SQ_INSTS_VALU"] / b["GRBM_GUI_ACTIVE"], 3)
        out["families"][fam] = row
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:4])
