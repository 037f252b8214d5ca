import sys, os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_gpu_fullsize_parity as T
r = T._run_case("mri_deepsup")
print({k: v for k, v in r.items() if k in ("e_lg","e_w","e_ce","e_dl","med","worst","wb","e_bn")}, r["l2s"][r["worst"]], r["bias"][r["wb"]])
top = sorted(r["l2s"].items(), key=lambda kv: -kv[1])[:8]
print(top)
