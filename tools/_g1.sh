python tools/bench_conv.py --c 32 --size 128 --iters 10 --profile 2>&1 | grep "wbf_gemm"
python tools/bench_conv.py --c 32 --size 128 --iters 10 --profile --opt wbf_variant=6 2>&1 | grep "wbf_gemm"
python tools/bench_conv.py --c 32 --size 64 --iters 10 --profile 2>&1 | grep "wbf_gemm"
python tools/bench_conv.py --c 32 --size 64 --iters 10 --profile --opt wbf_variant=6 2>&1 | grep "wbf_gemm"
python - <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import *
from test_gpu_ops import _desc
from oracle import vnet_numpy as O
d = dev()
rng = np.random.default_rng(0)
x = rng.standard_normal((2, 32, 16, 32, 16)).astype(np.float32)
w = (rng.standard_normal((32, 32, 5, 5, 5)) / 60).astype(np.float32)
ref = O.conv3d(x.astype(np.float64), w.astype(np.float64), None, (1,1,1), (2,2,2))
for v in (-1, 6):
    d.set_option("wbf_variant", v)
    yt = t_empty(2, 32, 16, 32, 16, fill=7.0)
    d.call("msk_conv3d_fwd", _desc((5,5,5),(1,1,1),(2,2,2)), t_from_ncdhw(x).msk(), vp(vec(w.ravel())), None, yt.msk())
    print("variant", v, "rel err %.2e" % rel_err(t_to_ncdhw(yt), ref))
PY
