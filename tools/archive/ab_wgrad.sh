for v in old lb3 lb0 lb3p lb0p; do
  for cfg in "32 128" "64 64" "128 32" "256 16"; do set -- $cfg
    MSEGK_LIB=medicalseg_amd/lib/ab/libmsegk_$v.so python tools/bench_conv.py --c $1 --size $2 --iters 10 --profile 2>&1 | grep -E "wbf_wgrad_h2_k" | head -1 | sed "s/^/$v c=$1 s=$2: /"
  done
done
