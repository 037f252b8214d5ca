cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/r18.txt
for cap in 2 4 8 16 32; do for rc in 4 8 16; do
  echo "== ew_cap=$cap reduce_cap=$rc" >> gpurun_out/r18.txt
  timeout 300 python tools/bench_elementwise.py ew_cap=$cap reduce_cap=$rc 2>&1 | grep "128^3x32\|128^3x16" >> gpurun_out/r18.txt
done; done
