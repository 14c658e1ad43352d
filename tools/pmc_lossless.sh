#!/bin/bash
# PMC counters of the VP8L chain (tools/lossless_times.py), one rocprofv3 run per pass.
# usage: [PMC_LIST=file-with-one-pass-per-line] tools/pmc_lossless.sh <tag> [images] [stream]
set -u
tag=$1; n=${2:-1024}; st=${3:-}
export TMPDIR=/tmp
out=gpurun_out/pmcl_$tag
mkdir -p $out
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -o p$i -- python3 tools/lossless_times.py $n $st > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
done < <(if [ -n "${PMC_LIST:-}" ]; then cat "$PMC_LIST"; else cat <<'LIST'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS
LIST
fi)
python3 tools/pmc_summary.py $out
