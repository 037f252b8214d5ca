for v in "$@"; do
  lib=medicalseg_amd/lib/ab/libmsegk_$v.so; [ $v = default ] && lib=medicalseg_amd/lib/libmsegk.so
  MSEGK_LIB=$lib python tools/bench_conv.py --c 32 --cn 3 --size 128 --iters 10 --profile 2>&1 | grep -E "conv_tk|conv_foldn|wgrad_cbs| ms " | sed "s/^/$v: /"
done
