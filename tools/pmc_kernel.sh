#!/bin/bash
# PMC snapshot of one kernel: bash tools/pmc_kernel.sh <kernel-name-substring> -- <command...>   (run on the GPU box)
KERN=$1; shift; shift
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  d=$R/gpurun_out/pmck_$(echo $set | md5sum | cut -c1-6); rm -rf $d; mkdir -p $d
  (cd $R && rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o v -- "$@" > /dev/null 2> $d/err.log)
  find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \;
  python3 - "$d" "$KERN" <<'PY'
import csv,sys,collections,glob
d,kern=sys.argv[1],sys.argv[2]
fs=glob.glob(d+'/*counter_collection.csv')
if not fs: print("no counters", open(d+'/err.log').read()[-300:]); sys.exit()
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if kern in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for c,v in acc.items(): print("%-32s %.4g per launch (%d launches)"%(c, v/n[c], n[c]))
PY
done
