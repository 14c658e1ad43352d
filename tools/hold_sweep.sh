#!/bin/bash
# Pipelined bench step against the finishing pass's hold-off and the number of workspaces: tools/hold_sweep.sh lib.so
lib=${1:-webp_amd/libwebp_amd.so}
for ws in 2 3; do for h in 0 10 20 40 80; do
  WEBP_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 5 --finish-delay-us $h --workspaces $ws 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('workspaces $ws hold $h us: step %.3f ms  stages %s' % (d['ms_per_step'], d['roofline']['stages_ms']))" || exit 1
done; done
