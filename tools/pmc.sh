#!/bin/bash
# Collects PMC counters for the bench kernels, one rocprofv3 run per pass (counter groups must fit the hardware).
# usage: [PMC_LIST=file-with-one-pass-per-line] tools/pmc.sh <tag> [bench args...]
set -u
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python3 bench.py --no-cpu-baseline "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done < <(if [ -n "${PMC_LIST:-}" ]; then cat "$PMC_LIST"; else cat <<'LIST'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
LIST
fi)
find $out -name "*counter_collection.csv" | head
