R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*" | sort -u | head -150 > $R/gpurun_out/counters.txt
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TCC_BUSY_avr"; do
  d=$R/gpurun_out/pmcg_$(echo $set | md5sum | cut -c1-6); rm -rf $d; mkdir -p $d
  (cd $R && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o v -- python tools/bench_conv.py --c 32 --size 128 --iters 2 > /dev/null 2> $d/err.log)
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python3 - "$d" wbf_gemm_k <<'PY'
import csv,sys,collections,glob
d,kern=sys.argv[1],sys.argv[2]
fs=glob.glob(d+'/*counter_collection.csv')
if not fs: print("no counters", open(d+'/err.log').read()[-400:]); sys.exit()
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if kern in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for c,v in acc.items(): print("%-32s %.4g per launch (%d launches)"%(c, v/n[c], n[c]))
PY
done
