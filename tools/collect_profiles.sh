#!/bin/bash
# gpurun_out/ (scratch; remove the pmcb_* / prof_* directories of earlier runs first: pmc_summarize.py reads every csv it finds)
# -> profiles/<round>/ (tracked): kernel stats of the C2 / C3 bench commands, the bench lines, the PMC csv + summary
rd=${1:-r02}
mkdir -p profiles/$rd
for c in C2 C3; do
  f=$(ls -t $(find gpurun_out/prof_$c -name "*kernel_stats.csv") 2>/dev/null | head -1)   # newest run
  [ -n "$f" ] && cp "$f" profiles/$rd/${c}_kernel_stats.csv
done
for c in C2 C3 C5; do [ -s gpurun_out/bench_$c.json ] && cp gpurun_out/bench_$c.json profiles/$rd/${c}_bench.json; done
for t in fetch write sq grbm; do
  f=$(ls -t $(find gpurun_out/pmcb_$t -name "*counter_collection.csv") 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" profiles/$rd/pmc_${t}_counter_collection.csv
done
python tools/pmc_summarize.py gpurun_out profiles/$rd/pmc_summary.json C2
ls -la profiles/$rd
