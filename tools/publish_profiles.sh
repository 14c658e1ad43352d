#!/bin/bash
# Copies what tools/collect_profiles.sh <tag> left under gpurun_out/prof_<tag>/ into profiles/<tag>_* (the judged set).
set -eu
tag=$1
src=gpurun_out/prof_$tag
for f in bench_1024.json bench_1024_serial.json kernel_stats_pipelined.txt kernel_stats_serial.txt pmc_sq_summary.txt pmc_lds_summary.txt \
         alpha_kernel_stats.txt lossless_kernel_stats.txt resolve_time.txt; do
  cp $src/$f profiles/${tag}_$f
done
cp $(find $src/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) profiles/${tag}_pmc_fetch_size.csv
cp $(find $src/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) profiles/${tag}_pmc_write_size.csv
cp $(find $src/pmc_sq -name "*counter_collection.csv" | head -1) profiles/${tag}_pmc_sq_insts.csv
cp $(find $src/pmc_lds -name "*counter_collection.csv" | head -1) profiles/${tag}_pmc_lds.csv
cp $src/traffic.json profiles/r03_traffic.json
cp $src/resolve_time_2048.txt profiles/${tag}_resolve_time_2048.txt
ls -la profiles/${tag}_*
