TAG=r04r; OUT=gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE WRITE_SIZE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_avr TCC_TAG_STALL_sum"; do
  i=$((i+1))
  ( cd /tmp && RW_LAYERS=layer16 timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$R/$OUT/p$i" -o pmc -- \
      python "$R/scripts/dconv_bench.py" > "$R/$OUT/p$i.log" 2>&1 ); echo "pass $i exit $?"
done
python - <<PY
import csv, glob, json, re
acc = {}
for path in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = re.sub(r'\(.*$', '', re.sub(r'^void ', '', row['Kernel_Name']))
        if ('dconv' not in k and 'wino36h' not in k) or 'pack' in k: continue
        k = k + ' grid=' + row.get('Grid_Size', '?')
        e = acc.setdefault(k, {}).setdefault(row['Counter_Name'], [0.0, set()])
        e[0] += float(row['Counter_Value']); e[1].add((path, row['Dispatch_Id']))
out = {k: {n: v[0] / max(len(v[1]), 1) for n, v in c.items()} for k, c in acc.items()}
json.dump(out, open("$OUT/pmc_mem_summary.json", 'w'), indent=1, sort_keys=True)
for k, v in sorted(out.items()): print(k, json.dumps(v, sort_keys=True))
PY
tail -3 $OUT/p4.log
rm -rf $OUT/p[0-9]
