for v in 1 2 3 4; do
  echo "== BPF=$v"
  for cs in "32 128" "64 64" "32 64"; do set -- $cs
    MSEGK_LIB=medicalseg_amd/lib/ab/libmsegk_bpf$v.so python tools/bench_conv.py --c $1 --size $2 --iters 10 --profile 2>&1 | grep -E "wbf_gemm" | head -2 | tr '\n' ' '; echo " c=$1 s=$2"
  done
done
