"""Launches of a rocprofv3 --kernel-trace CSV that leave CUs idle: per (kernel, workgroups) the calls per step and the time,
for launches below --max-wgs workgroups and above --min-us microseconds.  Use a one-stream run (bench.py --opt wgrad_async=0)
so that a kernel's duration is its own.

    python tools/grid_scan.py gpurun_out/prof_serial/r01_kernel_trace.csv --steps 4"""
import argparse
import collections
import csv
import re


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*$", "", n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--max-wgs", type=int, default=768)
    ap.add_argument("--min-us", type=float, default=15.0)
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(a.trace)):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        k = (short(r["Kernel_Name"]), g // w, w)
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = [(v[1] / a.steps, k, v[0] / a.steps) for k, v in agg.items() if k[1] < a.max_wgs and v[1] / v[0] > a.min_us]
    out.sort(reverse=True)
    for t, k, c in out[:60]:
        print("%-52s wgs %5d x %4d thr  calls/step %5.1f  us/step %8.1f  avg %7.1f us" % (k[0][:52], k[1], k[2], c, t, t / c))


if __name__ == "__main__":
    main()
