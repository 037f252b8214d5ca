"""Ordered per-launch listing of ONE training step from a rocprofv3 --kernel-trace csv of bench.py:
   python tools/step_trace.py <kernel_trace.csv> [step_index_from_end=1]
One line per launch in start order: start offset (us), duration (us), gap to the previous kernel of the same queue (us),
queue, workgroups, short kernel name.  Then per-kernel totals and the sub-40-us census the review asks for."""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), gx // max(1, wg)))
rows.sort()
# a step starts with in_tr's forward convolution (one launch per step; the eager optimizer re-packs per block, so the pack is no cut)
cuts = [i for i, r in enumerate(rows) if "conv_c1_" in r[2] and "wgrad" not in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = cuts[-back - 1], cuts[-back]
step = rows[lo:hi]
t0 = step[0][0]
last_end = {}
qn = {}
print("# step: %d launches, %.3f ms" % (len(step), (step[-1][1] - t0) / 1e6))
print("# start_us  dur_us  gap_us  q  workgroups  kernel")
tot = defaultdict(lambda: [0, 0.0])
for s, e, n, q, wgs in step:
    qi = qn.setdefault(q, len(qn))
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print("%9.1f %7.1f %6.1f  %d %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, qi, wgs, short(n)))
    k = short(n).split("<")[0]
    tot[k][0] += 1
    tot[k][1] += (e - s) / 1e3
print("# per kernel: launches, total us")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("#  %-44s %4d %9.1f" % (k, c, t))
small = [(e - s) / 1e3 for s, e, n, q, w in step if (e - s) < 40000]
print("# launches under 40 us: %d, %.3f ms; under 20 us: %d, %.3f ms" % (
    len(small), sum(small) / 1e3, len([x for x in small if x < 20]), sum(x for x in small if x < 20) / 1e3))
