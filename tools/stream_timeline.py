"""Two-stream picture of ONE training step from a rocprofv3 --kernel-trace csv of bench.py (default launch mode):
   python tools/stream_timeline.py <kernel_trace.csv> [step_index_from_end=1]
Steps are cut at in_tr's forward convolution (conv_c1_*, one launch per step).  Per hardware queue: busy time, the kernels' own time; then how long
both queues are busy at once, how long the side queue runs ALONE (exposed weight-gradient tail) and which compute-queue
kernels the side queue's kernels ran next to (by overlap time)."""
import csv
import sys
from collections import defaultdict

def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("<")[0].split("(")[0][-40:]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# a step starts with in_tr's forward convolution (one launch per step; round 5: the eager optimizer re-packs the weights per block,
# so the re-pack no longer marks the end of a step)
cuts = [i for i, r in enumerate(rows) if "conv_c1_" in r[2] and "wgrad" not in r[2]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = cuts[-back - 1], cuts[-back]
step = rows[lo:hi]
t0, t1 = step[0][0], step[-1][1]
print("step: %d kernels, %.3f ms" % (len(step), (t1 - t0) / 1e6))
byq = defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _ in byq[q]))
U = {q: union([(s, e) for s, e, _ in byq[q]]) for q in qs}
for q in qs:
    print("queue %s: %d kernels, busy %.3f ms (sum of kernel times %.3f)" % (q, len(byq[q]), sum(e - s for s, e in U[q]) / 1e6,
                                                                          sum(e - s for s, e, _ in byq[q]) / 1e6))
if len(qs) >= 2:
    main, side = qs[0], qs[1]
    both = inter(U[main], U[side])
    side_busy = sum(e - s for s, e in U[side])
    print("both busy %.3f ms; side alone %.3f ms; main alone %.3f ms; idle %.3f ms" % (
        both / 1e6, (side_busy - both) / 1e6, (sum(e - s for s, e in U[main]) - both) / 1e6,
        ((t1 - t0) - sum(e - s for s, e in union([(s, e) for s, e, _, _ in step]))) / 1e6))
    # which main-queue kernels do the side kernels overlap with
    acc = defaultdict(float)
    for s, e, n in byq[side]:
        for ms, me, mn in byq[main]:
            o = min(e, me) - max(s, ms)
            if o > 0:
                acc[(short(n), short(mn))] += o
    print("side kernel  next to  main kernel: overlap ms")
    for (a, b), v in sorted(acc.items(), key=lambda kv: -kv[1])[:25]:
        print("  %-40s %-40s %.3f" % (a, b, v / 1e6))
    # exposed tail: time after the last main-queue kernel before the optimizer during which only the side queue runs
    opt_start = min(s for s, e, n in byq[main] if "sgd_momentum_k" in n and s > (t0 + t1) // 2)   # (the step opens with the previous step's late-gradient update)
    last_main_before_opt = max(e for s, e, n in byq[main] if e <= opt_start)
    print("last compute-queue kernel ends %.3f ms before the optimizer starts (side-queue tail + join)" % ((opt_start - last_main_before_opt) / 1e6))
    # alone-time of the side queue by kernel
    alone = defaultdict(float)
    for s, e, n in byq[side]:
        o = inter([[s, e]], U[main])
        alone[short(n)] += (e - s) - o
    print("side kernels running ALONE (no compute-queue kernel at the same time):")
    for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:10]:
        print("  %-40s %.3f ms" % (k, v / 1e6))

    # main-queue kernels: time while a side kernel is co-resident vs alone, per kernel class
    co = defaultdict(lambda: [0.0, 0.0, 0])
    for s, e, n in byq[main]:
        o = inter([[s, e]], U[side])
        c = co[short(n)]
        c[0] += o
        c[1] += (e - s) - o
        c[2] += 1
    print("compute-queue kernels: launches, total ms, of which next to a side kernel")
    for k, (a, b, cnt) in sorted(co.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:30]:
        print("  %-40s %4d %8.3f %8.3f" % (k, cnt, (a + b) / 1e6, a / 1e6))
