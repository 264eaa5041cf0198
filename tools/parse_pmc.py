#!/usr/bin/env python
"""Per-launch PMC summary of the library's kernels from the rocprofv3 passes of tools/collect_profiles.sh.

    python tools/parse_pmc.py <dir with fetch/ write/ sq/ sq2/ sub-directories> [kernel-name-substring ...]

One JSON object per kernel: average launch duration (kernel trace of the same pass), HBM bytes per launch
(FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half of the bytes of wide coalesced reads: FETCH x 2, per
/opt/skills/guides/MI355X_MICROARCH.md section HBM), MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8 XCDs), effective clock, instruction mix per MFMA, wait shares of the wave cycles."""
import csv
import glob
import json
import sys

DEFAULT = ["syrk_pre_kernel", "syrk_patch_kernel", "syrk_flat_kernel", "patch_prep_kernel", "unfold_prep_kernel", "syrk_reduce_kernel", "gemv_rows_kernel", "outer_update_kernel", "outer_update_dma_kernel", "outer_update_wide_kernel",
           "inner_update_kernel", "panel_product_kernel", "panel_product_wide_kernel", "chol_diag_kernel", "chol_panel_kernel",
           "supd32_kernel", "xrows32_kernel",
           "gemm_f32_kernel", "gemm_nt_kernel", "gemm_f64_kernel", "corr_prep_kernel", "corr_assemble_kernel", "jacobi_pair_kernel", "jacobi_rows_kernel", "jacobi_cols_kernel"]


def collect(d, kernel):
    files = glob.glob(d + "/*/*counter_collection.csv")
    if not files:
        return {}
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
    acc, n, t = {}, {}, []
    seen = set()
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"]
        if kernel + "(" not in name and not name.endswith(kernel) and ("::" + kernel) not in name:
            continue
        acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            t.append(dur.get(r["Dispatch_Id"], 0))
    if not t:
        return {}
    out = {k: acc[k] / n[k] for k in acc}
    out["_launches"] = len(t)
    out["_avg_ns"] = sum(t) / max(len(t), 1)
    return out


def summarise(root, kernel):
    fetch, write = collect(root + "/fetch", kernel), collect(root + "/write", kernel)
    sq, sq2 = collect(root + "/sq", kernel), collect(root + "/sq2", kernel)
    if not (fetch or write or sq or sq2):
        return None
    res = {"kernel": "curv::" + kernel, "per": "launch (average over the profiled launches)"}
    if fetch and write:
        res["launches_profiled"] = fetch["_launches"]
        res["FETCH_SIZE_KiB_raw"] = fetch["FETCH_SIZE"]
        res["WRITE_SIZE_KiB_raw"] = write["WRITE_SIZE"]
        res["hbm_bytes_per_launch"] = (2.0 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024.0
        res["avg_kernel_ns_fetch_pass"] = fetch["_avg_ns"]
        res["hbm_GBps_fetch_pass"] = res["hbm_bytes_per_launch"] / max(fetch["_avg_ns"], 1.0)
    if sq:
        for k in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS",
                  "SQ_WAVE_CYCLES"):
            if k in sq:
                res[k] = sq[k]
        if sq.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in sq:
            cycles = sq["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
            res["clock_GHz"] = cycles / sq["_avg_ns"]
            res["mfma_pipe_utilisation"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles)   # 256 CUs x 4 SIMDs
        m = sq.get("SQ_INSTS_MFMA", 0.0)
        if m > 0:
            # SQ_INSTS_VALU counts the MFMAs as well
            res["valu_per_mfma"] = (sq.get("SQ_INSTS_VALU", 0.0) - m) / m
            res["salu_per_mfma"] = sq.get("SQ_INSTS_SALU", 0.0) / m
            res["lds_per_mfma"] = sq.get("SQ_INSTS_LDS", 0.0) / m
        res["avg_kernel_ns_sq_pass"] = sq["_avg_ns"]
    if sq2 and sq2.get("SQ_WAVE_CYCLES"):
        w = sq2["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if k in sq2:
                res[k.lower() + "_share"] = sq2[k] / w
        if sq2.get("SQ_LDS_IDX_ACTIVE"):
            res["lds_bank_conflict_share"] = sq2.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq2["SQ_LDS_IDX_ACTIVE"]
    return res


def main():
    root = sys.argv[1]
    args = sys.argv[2:]
    if args and args[0] == "--factor-build":
        # bench.py's `roofline.traffic`: HBM bytes of ONE update(), everything inside the timed window: the padding /
        # pre-tiling passes, the three MFMA kernels, the k-slice reductions and the 3x3 assembly.  Per kernel: average
        # bytes per launch x launches per update (= its launches / the launches of syrk_flat_kernel, one per update)
        names = ("corr_prep_kernel", "patch_prep_kernel", "unfold_prep_kernel", "syrk_pre_kernel", "syrk_patch_kernel",
                 "syrk_flat_kernel", "syrk_reduce_kernel", "corr_assemble_kernel")
        parts = {k: summarise(root, k) for k in names}
        parts = {k: v for k, v in parts.items() if v}
        updates = max(parts.get("syrk_flat_kernel", {}).get("launches_profiled", 1), 1)
        total, slab = 0.0, 0.0
        for k, v in parts.items():
            v["launches_per_update"] = v.get("launches_profiled", 0) / updates
            v["hbm_bytes_per_update"] = v.get("hbm_bytes_per_launch", 0.0) * v["launches_per_update"]
            total += v["hbm_bytes_per_update"]
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import syrk_source_sha16                  # bench.py reports the figure only for these very sources
        print(json.dumps({"source_sha16": syrk_source_sha16(),
                          "kernel": " + ".join("curv::" + k for k in parts),
                          "per": "update() (everything curv_kfac_accumulate_ex enqueues), average over the profiled updates",
                          "hbm_bytes_per_launch": total,
                          "mfma_kernels_bytes_per_update": sum(parts[k]["hbm_bytes_per_update"] for k in ("syrk_pre_kernel", "syrk_patch_kernel", "syrk_flat_kernel") if k in parts),
                          "reduce_pass_bytes_per_update": parts.get("syrk_reduce_kernel", {}).get("hbm_bytes_per_update", 0.0),
                          "parts": parts}, indent=1))
        return
    kernels = args or DEFAULT
    out = [s for s in (summarise(root, k) for k in kernels) if s]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
