export TMPDIR=/tmp RW_BATCH=64 RW_UP_ALGO=wino RW_LAYERS=${RW_LAYERS:-layer13}
R=$PWD
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -o p -- python $R/scripts/conv_bench.py > /tmp/pm.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: [0.0,set()])
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'up_wino' in r['Kernel_Name'] and 'pack' not in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1].add(r['Dispatch_Id'])
for k,v in sorted(acc.items()): print(k, v[0]/max(len(v[1]),1), len(v[1]))
PY
done
