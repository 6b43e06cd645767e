#!/bin/bash
# rocprofv3 kernel stats of the key-statistics sweep (config 4) and of the rank-1 edit (config 3)
OUT=gpurun_out/sweep_prof; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o sweep -- python "$R/bench.py" --workload sweep --steps 3 --warmup 1 > "$R/$OUT/sweep.json" 2> "$R/$OUT/sweep.err" ); echo "sweep exit $?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof_edit" -o edit -- python "$R/bench.py" --workload edit --steps 2 --warmup 1 > "$R/$OUT/edit.json" 2> "$R/$OUT/edit.err" ); echo "edit exit $?"
for n in sweep edit; do f=$(find $OUT -name "${n}_kernel_stats.csv" | head -1); echo "== $n"; head -9 "$f" | cut -c1-150; done
