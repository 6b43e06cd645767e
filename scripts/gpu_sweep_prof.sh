#!/bin/bash
OUT=gpurun_out/sweep_prof; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o sweep -- python "$R/bench.py" --workload sweep --steps 3 --warmup 1 > "$R/$OUT/sweep.json" 2> "$R/$OUT/sweep.err" ); echo "exit $?"
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-170
cat $OUT/sweep.json | cut -c1-600
