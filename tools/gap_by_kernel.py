"""Idle time before / after each kernel NAME in a rocprofv3 --kernel-trace csv of a serial run (--opt wgrad_async=0):
   python tools/gap_by_kernel.py <kernel_trace.csv>
For every kernel (second half of the trace = warm steps): launches, mean gap to the previous kernel's end, mean gap from its own end to
the next kernel's start.  Round 5: every wbf_gemm launch sits between two ~6 us gaps while every other pair is back to back."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


acc = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i in range(1, len(rows) - 1):
    s, e, n = rows[i]
    a = acc[short(n)]
    a[0] += 1
    a[1] += max(0, s - rows[i - 1][1]) / 1e3
    a[2] += max(0, rows[i + 1][0] - e) / 1e3
    a[3] += (e - s) / 1e3
tot = sum(max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)) / 1e3
print("# total idle %.1f us over %d launches" % (tot, len(rows)))
print("# launches  gap_before_us  gap_after_us  avg_dur_us  kernel")
for k, (c, b, a, d) in sorted(acc.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    if (b + a) / c < 0.5:
        continue
    print("%6d %8.2f %8.2f %10.1f  %s" % (c, b / c, a / c, d / c, k))
