#!/bin/bash
# gpurun_out/ (scratch; remove the pmcb_* / prof_* directories of earlier runs first: pmc_summarize.py reads every csv it finds)
# -> profiles/<round>/ (tracked): kernel stats of the bench commands, the bench lines, the PMC csv + summary per configuration
rd=${1:-r04}
mkdir -p profiles/$rd
for c in C2 C3 C5; do
  f=$(ls -t $(find gpurun_out/prof_$c -name "*kernel_stats.csv") 2>/dev/null | head -1)   # newest run
  [ -n "$f" ] && cp "$f" profiles/$rd/${c}_kernel_stats.csv
  [ -s gpurun_out/bench_$c.json ] && cp gpurun_out/bench_$c.json profiles/$rd/${c}_bench.json
  found=0
  for t in fetch write sq grbm; do
    f=$(ls -t $(find gpurun_out/pmcb_${c}_$t -name "*counter_collection.csv") 2>/dev/null | head -1)
    [ -n "$f" ] && cp "$f" profiles/$rd/pmc_${c}_${t}_counter_collection.csv && found=1
  done
  [ $found = 1 ] && python tools/pmc_summarize.py gpurun_out profiles/$rd/pmc_summary_$c.json $c
done
f=$(ls -t $(find gpurun_out/prof_eval -name "*kernel_stats.csv") 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" profiles/$rd/eval_kernel_stats.csv
[ -s gpurun_out/march_walk_timing.txt ] && cp gpurun_out/march_walk_timing.txt profiles/$rd/
ls -la profiles/$rd
