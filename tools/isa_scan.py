"""Scan the gfx950 ISA of the HIP sources for loops whose global loads are waited for one by one.

Two pathologies cost this project measurable time and both look the same in the ISA -- a load followed directly by
`s_waitcnt vmcnt(0)` inside an inner loop:
  * `x = cond ? f(load(p)) : 0` -- the compiler sinks the load into a branch on `cond`, so a batch of loads meant to be in
    flight together becomes one round trip per element (wgrad_c1_mfma_k<5, true>: 0.41 instead of 0.24 ms);
  * a dependent accumulation `for z: acc += p[z * stride]` -- one load in flight (wbf_tout_k, wbf_wgrad_reduce_k split-K slabs).
Also: `(half)(float)double` folds into a double -> half conversion, which has no instruction (~40 integer operations;
wbf_pack_weights_k) -- look for kernels without v_cvt_f16_f32 where one is expected.

    python tools/isa_scan.py [kernel-name-substring ...]     (compiles medicalseg_amd/csrc/*.hip to /tmp/msegk_isa/*.s first)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/msegk_isa"


def compile_all():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "medicalseg_amd/csrc/*.hip"))):
        dst = os.path.join(OUT, os.path.basename(src)[:-4] + ".s")
        if os.path.exists(dst) and os.path.getmtime(dst) > os.path.getmtime(src):
            continue
        procs.append(subprocess.Popen(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                       "-I" + os.path.join(ROOT, "medicalseg_amd/csrc"), "-Wno-unused-value", "-Wno-comment", "-S",
                                       "--cuda-device-only", "-o", dst, src], stderr=subprocess.DEVNULL))
    for p in procs:
        p.wait()


def main():
    want = sys.argv[1:]
    compile_all()
    for f in sorted(glob.glob(os.path.join(OUT, "*.s"))):
        txt = open(f).read().split("\n")
        i = 0
        while i < len(txt):
            m = re.match(r"^(_ZN[\w]+):\s", txt[i])
            if not m:
                i += 1
                continue
            j = i
            while j < len(txt) and "s_endpgm" not in txt[j]:
                j += 1
            body = txt[i:j]
            i = j + 1
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
            if want and not any(w in name for w in want):
                continue
            labels = {mm.group(1): k for k, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
            rows = []
            for k, l in enumerate(body):
                mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
                if not (mm and mm.group(1) in labels and labels[mm.group(1)] < k):
                    continue
                a = labels[mm.group(1)]
                seg = body[a:k]
                inner = not any((m2 := re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ll)) and m2.group(1) in labels
                                and a <= labels[m2.group(1)] < a + kk for kk, ll in enumerate(seg))
                loads = sum(1 for x in seg if re.search(r"\b(buffer_load|global_load)", x))
                waits = sum(1 for x in seg if re.search(r"s_waitcnt.*vmcnt\(0\)", x))
                if loads and inner:
                    rows.append((k - a, loads, waits, sum("v_mfma" in x for x in seg)))
            if rows:
                print(name[:70])
                for n, lo, wa, mf in rows:
                    print("     inner loop of %4d lines: %3d loads, %2d vmcnt(0) waits, %3d MFMAs%s"
                          % (n, lo, wa, mf, "   <-- every load waited for" if wa >= lo else ""))


if __name__ == "__main__":
    main()
