set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wbf.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r1_wbf_tests.txt
for fz in 1 0; do
 for cfg in "32 128" "64 64" "32 64" "64 32"; do set -- $cfg;
  timeout 300 python tools/bench_conv.py --c $1 --size $2 --opt wbf_fuse=$fz --profile 2>&1 | grep -v "^$" | sed "s/^/fuse=$fz /" >> gpurun_out/r1_conv_ab.txt
 done
done
timeout 600 python bench.py --no-cpu-baseline --skip-serialized > gpurun_out/r1_bench_fused.json 2> gpurun_out/r1_bench_fused.err
timeout 600 python bench.py --no-cpu-baseline --skip-serialized --opt wbf_fuse=0 > gpurun_out/r1_bench_unfused.json 2>> gpurun_out/r1_bench_fused.err
