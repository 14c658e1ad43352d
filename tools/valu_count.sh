#!/bin/bash
# Dynamic instruction counts per kernel (one rocprofv3 --pmc pass over `bench.py --serial --one-launch`), the metric the
# lossy kernels are tuned by: they are bound by VALU issue, so time follows SQ_INSTS_VALU.  usage: tools/valu_count.sh <tag>
tag=${1:-x}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $out/p1 -o p1 -- python3 bench.py --no-cpu-baseline --serial --steps 2 --warmup 1 > $out/p1.log 2>&1 || { echo "pass failed"; tail -5 $out/p1.log; exit 1; }
python3 tools/pmc_summary.py $out 2>/dev/null | grep -A9 "vp8_mb_kernel\|vp8_finish_kernel" | grep -E "==|INSTS_VALU|INSTS_SALU|INSTS_LDS|BUSY_CYCLES"
