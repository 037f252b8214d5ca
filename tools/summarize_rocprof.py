#!/usr/bin/env python
"""Turn rocprofv3 output directories (gpurun_out/prof_{stats,fetch,write}) into the small tracked
summaries under profiles/:  <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
<tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE per launch for the MFMA kernels, with the
gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md section HBM).

    python tools/summarize_rocprof.py r01
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                d[r["Kernel_Name"]][0] += float(r["Counter_Value"])
                d[r["Kernel_Name"]][1] += 1
    return d


BUCKETS = (("lu_gemm", ("wbf_gemm_k", "wbf_gemm_fused_k")),
           ("lu_wgrad", ("wbf_wgrad_k", "wbf_wgrad_reduce")),
           ("lu_transforms", ("wbf_tin_k", "wbf_tin_dual_k", "wbf_tout_k", "wbf_pack_", "absmax_k")),
           ("ks_convs", ("gconv_ks_fwd_k", "gconv_ks_lds_k", "gconv_kst_k", "convT_scatter_mfma_k", "convT_scatter_lds_k", "gconv_gather_mfma_k", "wgrad_ks_mfma_k", "wgrad_ks2_k", "wgrad_mfma_k")),
           ("tiny_channel", ("conv_foldn", "conv_tk_", "conv_halo_tightk", "wgrad_cbs", "conv_c1_", "wgrad_c1_", "wgrad_pw_small",
                             "pointwise_small", "conv_halo_valu", "pack_foldn", "pack_tk")),
           ("loss_optim", ("loss_", "sgd_momentum_k", "adam_k", "class_weights")),
           ("bn_prelu_join", ("affine_act", "bn_", "sums_merge_k", "param_grads_k", "copy_scale_k", "dropout_mask_k",
                              "channel_sum", "bias_grad")),
           ("weight_packs_reduces", ("pack_weights_k", "small_pack_", "pack_scatter", "wgrad_reduce", "wgrad_prereduce")))


def bucket_of(name):
    for b, keys in BUCKETS:
        if any(name.startswith(k) or ("<" in name and name.split("<")[0].startswith(k)) for k in keys):
            return b
    return "other"


def main(tag):
    go = os.path.join(ROOT, "gpurun_out")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = os.path.join(go, "prof_stats", "r01_kernel_stats.csv")
    if not os.path.exists(stats):
        stats = os.path.join(go, "prof_stats", tag + "_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(out, f"{tag}_rocprofv3_kernel_stats.csv"))
        bj = os.path.join(go, "prof_stats", "bench.json")
        if os.path.exists(bj):
            shutil.copy(bj, os.path.join(out, f"{tag}_rocprofv3_kernel_stats_bench_line.json"))
    fe_p = os.path.join(go, "prof_fetch", "r01_counter_collection.csv")
    wr_p = os.path.join(go, "prof_write", "r01_counter_collection.csv")
    if os.path.exists(fe_p) and os.path.exists(wr_p):
        fe, wr = agg(fe_p, "FETCH_SIZE"), agg(wr_p, "WRITE_SIZE")
        import datetime
        res = {"_captured": datetime.date.today().isoformat() + " (" + tag + ")",
               "_note": "per-launch averages over one training step; FETCH_SIZE/WRITE_SIZE are in KiB; "
                        "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 rocprofv3 reports half of a wide "
                        "coalesced read stream, MI355X_MICROARCH.md section HBM; WRITE_SIZE uncalibrated)"}
        steps = int(os.environ.get("PROFILE_STEPS", "4"))      # training steps inside the PMC passes (profile_gpu.sh: 3 warm-up + 1)
        total = 0.0
        buckets = collections.defaultdict(float)
        for k in sorted(set(fe) | set(wr)):     # EVERY kernel of the step (round 2 listed the convolution families only)
            n = max(fe[k][1], wr[k][1], 1)
            f_kb, w_kb = fe[k][0] / max(fe[k][1], 1), wr[k][0] / max(wr[k][1], 1)
            short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
            per_launch = (2 * f_kb + w_kb) * 1024
            res[short] = {"launches": n, "FETCH_SIZE_KiB": round(f_kb, 1), "WRITE_SIZE_KiB": round(w_kb, 1),
                          "hbm_bytes_per_launch": int(per_launch), "bucket": bucket_of(short)}
            total += per_launch * n / steps
            buckets[bucket_of(short)] += per_launch * n / steps
        res["_whole_step"] = {"hbm_bytes_per_step": int(total), "steps_profiled": steps,
                              "buckets_bytes_per_step": {b: int(v) for b, v in sorted(buckets.items())},
                              "note": "sum over ALL kernel launches of one training step (VNet 128^3, batch 2); memset / copy "
                                      "commands of the runtime are not kernels and are not counted"}
        with open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    sq_p = os.path.join(go, "prof_sq", "r01_counter_collection.csv")
    if os.path.exists(sq_p):
        # MFMA-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs,
        # GRBM_GUI_ACTIVE over the 8 XCDs (pass run with wgrad_async=0 so that kernels do not overlap)
        tot = collections.defaultdict(lambda: collections.defaultdict(float))
        with open(sq_p) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        res = {"_note": "mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), summed over "
                        "all launches of one training step, weight gradients on the main stream (--opt wgrad_async=0)"}
        for k, v in tot.items():
            if any(t in k for t in ("mfma", "wino", "tightk", "wbf_", "foldn", "gconv_ks", "wgrad_cbs", "conv_tk")) and v.get("GRBM_GUI_ACTIVE"):
                res[k] = {"mfma_busy_frac": round((v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (v["GRBM_GUI_ACTIVE"] / 8), 4),
                          "gui_active_cycles_per_xcd": int(v["GRBM_GUI_ACTIVE"] / 8)}
        with open(os.path.join(out, f"{tag}_mfma_busy.json"), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    print("wrote summaries to", out)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
