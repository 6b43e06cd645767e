#!/bin/bash
# Copies the summaries of one `scripts/gpu_final.sh <tag>` run from gpurun_out/ (scratch) into profiles/ (tracked).
T="${1:?tag}"; S=gpurun_out/$T; D=profiles
cp $S/bench.json $D/${T}_bench1024_b64.json
f=$(find $S/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/${T}_bench1024_b64_kernel_stats.csv
cp $S/kernel_stats_steady.csv $D/${T}_bench1024_b64_kernel_stats_steady.csv
cp $S/pmc_summary.json $D/${T}_pmc_summary_b64.json
cp $S/pmc_mfma_summary.json $D/${T}_pmc_mfma_summary_b64.json
cp $S/edit.json $D/${T}_bench_edit.json
cp $S/watermark.json $D/${T}_bench_watermark.json
cp $S/cb_all.json $D/${T}_conv_bench_b64.json
cp $S/solve_probe.json $D/${T}_solve_probe.json
cp $S/micro_probe.json $D/${T}_micro_probe.json
cp $S/pytest_gpu.log $D/${T}_pytest_gpu.log
cp $S/sweep_prof/sweep.json $D/${T}_bench_sweep_under_rocprof.json 2>/dev/null
f=$(find $S/sweep_prof -name "sweep_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/${T}_sweep1024_l8_kernel_stats.csv
f=$(find $S/sweep_prof -name "edit_kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $D/${T}_edit_horse256_kernel_stats.csv
for n in fullsize_b64_gen_s1024_full_fuse0 fullsize_b64_gen_s1024_full_fuse1 fullsize_b64_gen_s256_full_fuse1 fullsize_edit_parity \
         fullsize_sweep_parity fullsize_watermark_parity two_layer_gradients solve_parity_two_layer_hook0; do
  [ -f gpurun_out/$n.json ] && cp gpurun_out/$n.json $D/${T}_$n.json
done
ls $D | grep "^${T}_" | wc -l
