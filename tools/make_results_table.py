#!/usr/bin/env python
"""Regenerate the results block of README.md (between the `<!-- results:begin -->` / `<!-- results:end -->` markers) from the
tracked evidence files of a round -- no hand-edited rows:
    python tools/make_results_table.py [r05]
Sources: profiles/<tag>_bench_line_default.json (bench.py's line), _bench_hip_events_serial_shapes.tsv (per-kernel HIP events,
one stream), _rocprofv3_kernel_stats.csv, _hbm_traffic.json, _mri_workload.json, _mri_deepsup_workload.json,
_unet3d_workload.json, _preprocess.json."""
import csv
import json
import os
import re
import sys


def _event_steps(r, b):
    """timed steps whose launches of the roofline kernel carried events (bench.py --roofline-every)"""
    if "event_steps" in r:
        return max(1, int(r["event_steps"]))
    import re
    m = re.search(r"in (\d+) of the (\d+) timed steps", r.get("events", ""))
    return int(m.group(1)) if m else b["steps"]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (TAG, name))


def jload(name):
    try:
        txt = open(P(name)).read().strip()
        return json.loads(txt.splitlines()[-1]) if txt.startswith('{"metric"') else json.loads(txt)
    except (OSError, ValueError):
        return None


rows = []
b = jload("bench_line_default.json")
if b:
    r = b["roofline"]
    rows.append(("training step, VNet 128^3 fp32, batch 2, 1 GPU (`python bench.py`; BASELINE configs[1])",
                 "**%.2f ms/step = %.1f M voxels/s** (%d steps after %d warm-up; synthetic CT volumes resident in HBM)"
                 % (b["ms_per_step"], b["value"] / 1e6, b["steps"], b["warmup"])))
    if b.get("ms_per_step_sustained"):
        rows.append(("the same step sustained over 150 further steps (untimed extra pass); in the plain optimizer order (what a rank of an N > 1 job runs)",
                     "%.2f ms/step = %.1f M voxels/s; %s" % (b["ms_per_step_sustained"], b["value_sustained"] / 1e6,
                                                            ("%.2f ms/step" % b["ms_per_step_plain_order"]) if b.get("ms_per_step_plain_order") else "n/a")))
    sf = r.get("strict_fp32")
    if sf:
        rows.append(("the same step with exact fp32 operands (`conv_split` 3, `roofline.strict_fp32`)",
                     "%.2f ms/step = %.1f M voxels/s" % (sf["ms_per_step"], sf["value"] / 1e6)))
    rows.append(("dominant kernel `%s` (matrix stage of every LUConv forward / data gradient, %d launches per step)" % (r["kernel"], r["launches"] // _event_steps(r, b)),
                 "%.4f ms per launch (HIP events in the step) = %.0f TFLOP/s executed on the 16-bit pipe = **%.3f of the %.0f TFLOP/s dense peak** "
                 "(%.0f TFLOP/s algorithmic = %s of it); one stream: %.4f ms = %.3f" % (r["avg_launch_ms"], r["achieved"], r["frac"], r["peak"], r["algorithmic_tflops"],
                                                                            ("%.3f" % r["frac_algorithmic"]) if "frac_algorithmic" in r else "?",
                                                                            r["serialized"]["avg_launch_ms"], r["serialized"]["frac"]) if r.get("serialized") else ""))
    if r.get("traffic"):
        rows.append(("HBM traffic of that kernel (PMC counters of the committed passes) vs its algorithmic bytes",
                     "%.1f MB per launch vs %.1f MB = %.2fx" % (r["traffic"] / 1e6, r["algorithmic_bytes_per_launch"] / 1e6, r["traffic"] / r["algorithmic_bytes_per_launch"])))
    rows.append(("whole step: executed matrix work at the hardware peaks / step time (`roofline.step_executed_frac`)", "%.3f" % r["step_executed_frac"]))
    h = r.get("hbm")
    if h:
        # the counters of the committed passes (the bench line quotes the file that was in profiles/ when it ran)
        tj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_hbm_traffic.json" % TAG)
        if os.path.exists(tj):
            ws = json.load(open(tj)).get("_whole_step")
            if ws:
                h = dict(h, counter_GB_per_step=ws["hbm_bytes_per_step"] / 1e9, ratio=ws["hbm_bytes_per_step"] / 1e9 / h["algorithmic_GB_per_step"],
                         source="profiles/%s_hbm_traffic.json:" % TAG)
        rows.append(("HBM traffic of one step (PMC counters FETCH_SIZE / WRITE_SIZE over all kernels) vs algorithmic bytes",
                     "%.1f GB vs %.2f GB = **%.2fx** (%s)" % (h["counter_GB_per_step"], h["algorithmic_GB_per_step"], h["ratio"], h["source"].split(":")[0])))
    c = b.get("cpu_baseline")
    if c:
        rows.append(("CPU baseline (torch-CPU / oneDNN restatement of the same step, kind `%s`)" % c["kind"],
                     "%.2f M voxels/s on %d threads of %s (%d logical CPUs); %s" % (c["value"] / 1e6, c["threads"], c.get("cpu_model", "?"), c.get("nproc", 0), c["sample"].split(": ")[-1])))
try:
    n = t = a = bb = 0.0
    with open(P("bench_hip_events_serial_shapes.tsv")) as f:
        steps = int(re.search(r"\((\d+) steps\)", f.readline()).group(1))
        for line in f:
            if line.startswith("#"):
                continue
            _, calls, total, avg, _ = line.rstrip("\n").split("\t")
            n += int(calls) / steps
            t += float(total) / steps
            if float(avg) < 0.040:
                a += int(calls) / steps
                bb += float(total) / steps
    rows.append(("launches per step (HIP-event tags, weight gradients on the compute stream)",
                 "%d launches, %.2f ms of kernel time; %d of them under 40 us = %.2f ms.  A dependent empty launch costs 2.2 us in the step (HISTORY.md section 4, round 5)" % (n, t, a, bb)))
except (OSError, AttributeError):
    pass
try:
    tt = nn = 0
    for r_ in csv.DictReader(open(P("rocprofv3_kernel_stats.csv"))):
        if "wbf_gemm_k<" in r_["Name"] or "wbf_gemm_fused_k<" in r_["Name"]:
            nn += int(r_["Calls"])
            tt += float(r_["TotalDurationNs"])
    if nn:
        rows.append(("the same kernel under `rocprofv3 --kernel-trace --stats` of the bench command", "%d launches, %.4f ms per launch" % (nn, tt / nn / 1e6)))
except OSError:
    pass
for name, label in (("mri_workload.json", "MRI VNet 512x512x12, 20 classes, batch 1 (BASELINE configs[4])"),
                    ("mri_deepsup_workload.json", "MRI VNetDeepSup, same shape"),
                    ("unet3d_workload.json", "UNet3D 192x192x64 fp16 matrix operands, batch 2 (builder-defined, BASELINE configs[3])")):
    w = jload(name)
    if not w:
        continue
    txt = "%.2f ms/step = %.1f M voxels/s, step_executed_frac %.3f" % (w["ms_per_step"], w["voxels_per_s"] / 1e6, w["roofline"]["frac"])
    il = w.get("inloop_preprocess")
    if il:
        txt += "; **with the in-loop preprocessing** (raw %s in pinned memory -> H2D -> normalize -> resample -> max-normalise on a second stream, one sample ahead): %.2f ms/step (resident input %.2f; the preprocessing alone %.2f ms per sample)" % (
            "x".join(str(v) for v in il["raw_shape"]), il["ms_per_step_with_inloop_preprocess"], il["ms_per_step_resident_input"], il["preprocess_alone_ms_per_sample"])
    rows.append((label, txt))
pp = jload("preprocess.json")
if pp:
    rows.append(("preprocessing kernels (HIP events, resident data; HBM streaming rate %.1f TB/s)" % (pp["hbm_streaming_GBps"] / 1e3),
                 "; ".join("%s %.1f GB/s" % (r_["kernel"].split(" (")[0], r_["GBps"]) for r_ in pp["rows"] if "GBps" in r_)))
out = ["<!-- results:begin (generated by tools/make_results_table.py %s from profiles/%s_*; do not edit by hand) -->" % (TAG, TAG), "",
       "| measured (round %s, MI355X) | |" % TAG.lstrip("r0"), "|---|---|"]
out += ["| %s | %s |" % row for row in rows]
out += ["", "<!-- results:end -->"]
block = "\n".join(out)
path = os.path.join(ROOT, "README.md")
s = open(path).read()
if "<!-- results:begin" in s:
    s = re.sub(r"<!-- results:begin.*?<!-- results:end -->", lambda m: block, s, flags=re.S)
else:
    raise SystemExit("README.md has no results markers")
open(path, "w").write(s)
print(block)
