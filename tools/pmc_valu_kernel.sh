R=$(pwd); export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  d=$R/gpurun_out/pmcv_$(echo $set | md5sum | cut -c1-6); rm -rf $d; mkdir -p $d
  (cd $R && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o v -- python tools/bench_valu_cn.py > /dev/null 2> $d/err.log)
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python3 - "$d" <<'PY'
import csv,sys,collections,glob
d=sys.argv[1]
fs=glob.glob(d+'/*counter_collection.csv')
if not fs: print("no counters", open(d+'/err.log').read()[-400:]); sys.exit()
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k=r['Kernel_Name']
    if 'conv_halo_valu_k' in k and '32, 3>' in k:
        acc['k'][r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for c,v in acc['k'].items(): print(c, v/n[c])
PY
done
