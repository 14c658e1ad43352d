#!/bin/bash
# Runs, on the GPU box, every measurement the round's profiles/ set is made of and leaves the results under
# gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/<tag>_*).  usage: tools/collect_profiles.sh <tag>
# Every GPU step has its own time limit; the script stops at the first failure.
set -u
tag=${1:-rXX}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
step() { echo "== $*"; }
run() { timeout -k 10 "$@" || { echo "FAILED: $*"; exit 1; }; }

step "bench, pipelined (default) and one stream"
run 400 python3 bench.py > $out/bench_1024.json 2> $out/bench_1024.err
run 200 python3 bench.py --serial --no-secondary > $out/bench_1024_serial.json 2> $out/bench_1024_serial.err

step "rocprofv3 kernel trace of the same two commands (fewer steps)"
run 300 rocprofv3 --kernel-trace --stats -d $out/kt_pipe -o pipe -- python3 bench.py --no-cpu-baseline --no-secondary --steps 20 > $out/kt_pipe.log 2>&1
run 300 rocprofv3 --kernel-trace --stats -d $out/kt_serial -o serial -- python3 bench.py --no-cpu-baseline --no-secondary --serial --steps 20 > $out/kt_serial.log 2>&1
python3 tools/prof_summary.py $(find $out/kt_pipe -name "*results.db" | head -1) $out/kernel_stats_pipelined.txt > /dev/null
python3 tools/prof_summary.py $(find $out/kt_serial -name "*results.db" | head -1) $out/kernel_stats_serial.txt > /dev/null

step "PMC: HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes), instruction counts, LDS"
for c in FETCH_SIZE WRITE_SIZE; do
  run 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --no-cpu-baseline --no-secondary --serial --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1
done
run 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq -o p -- python3 bench.py --no-cpu-baseline --no-secondary --serial --steps 2 --warmup 1 > $out/pmc_sq.log 2>&1
run 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_lds -o p -- python3 bench.py --no-cpu-baseline --no-secondary --serial --steps 2 --warmup 1 > $out/pmc_lds.log 2>&1
python3 tools/traffic_json.py $out 1024 $out/traffic.json > /dev/null
python3 tools/pmc_summary.py $out/pmc_sq > $out/pmc_sq_summary.txt 2>&1
python3 tools/pmc_summary.py $out/pmc_lds > $out/pmc_lds_summary.txt 2>&1

step "secondary configurations: kernel traces"
run 300 rocprofv3 --kernel-trace --stats -d $out/kt_alpha -o alpha -- python3 tools/alpha_times.py > $out/kt_alpha.log 2>&1
python3 tools/prof_summary.py $(find $out/kt_alpha -name "*results.db" | head -1) $out/alpha_kernel_stats.txt > /dev/null
run 300 rocprofv3 --kernel-trace --stats -d $out/kt_lossless -o ll -- python3 tools/lossless_times.py 1024 > $out/kt_lossless.log 2>&1
python3 tools/prof_summary.py $(find $out/kt_lossless -name "*results.db" | head -1) $out/lossless_kernel_stats.txt > /dev/null
run 200 python3 tools/resolve_time.py 512 > $out/resolve_time.txt 2>&1
run 200 python3 tools/resolve_time.py 2048 > $out/resolve_time_2048.txt 2>&1
[ -x tools/ubench/valu_rate_fp ] && run 100 ./tools/ubench/valu_rate_fp > $out/ubench_valu_rate_fp.txt 2>&1
# the raw databases are large and not needed once summarised
find $out -name "*results.db" -delete
echo "done: $out"
