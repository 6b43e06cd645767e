#!/bin/bash
# rocprofv3 kernel stats of the watermark job's erase solve (2 x 2001 steps on whole 16 x 16 maps).
TAG=${1:-r04l}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
python scripts/erase_solve.py | tee $OUT/erase_solve.jsonl
( cd /tmp && REPS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o erase -- python "$R/scripts/erase_solve.py" > "$R/$OUT/prof.log" 2>&1 ); echo "rocprof exit $?"
F=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$F" ] && cp $F $OUT/erase_kernel_stats.csv && head -14 $F | cut -c1-160
find $OUT/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
