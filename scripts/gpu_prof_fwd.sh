#!/bin/bash
# rocprofv3 kernel stats of the default 1024 forward (batch 64); usage: gpu_prof_fwd.sh <tag> [ENV=VAL ...]
tag=${1:-r06k}; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o fwd -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-extra --no-cpu-baseline > $out/bench.out 2> $out/bench.err
tail -1 $out/bench.out | cut -c1-300
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv 2>/dev/null
head -16 $out/kernel_stats.csv | cut -c1-150
rm -rf $out/prof
