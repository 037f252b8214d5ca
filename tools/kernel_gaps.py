"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv (serial run: --opt wgrad_async=0):
   python tools/kernel_gaps.py <kernel_trace.csv> [skip_first_n_kernels]
Prints busy time, idle time and the histogram of gaps, i.e. what a hipGraph / fewer launches could still recover."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2     # default: second half (warm steps)
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
print("kernels %d  span %.3f ms  busy %.3f ms  idle %.3f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6,
      (span - busy) / 1e6, 100.0 * (span - busy) / span))
for lo, hi in ((0, 1), (1, 2), (2, 4), (4, 8), (8, 16), (16, 64), (64, 1 << 30)):
    sel = [g for g in gaps if lo * 1000 <= g < hi * 1000]
    print("  gap %3d-%-4s us: %5d  total %.3f ms" % (lo, hi if hi < 1 << 20 else "inf", len(sel), sum(sel) / 1e6))
big = sorted(((g, rows[i][2][:60], rows[i + 1][2][:60]) for i, g in enumerate(gaps)), reverse=True)[:8]
for g, a, b in big:
    print("  %.1f us after %s -> %s" % (g / 1e3, a, b))
