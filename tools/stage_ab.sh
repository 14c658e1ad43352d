#!/bin/bash
# Pipelined bench step per library build, with the stretched duration of each kernel (bench.py's stages_ms):
#   tools/stage_ab.sh lib1.so lib2.so ...      (run on the GPU box)
for lib in "$@"; do
  WEBP_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-44s step %.3f ms  stages %s' % ('$lib'.split('/')[-1], d['ms_per_step'], d['roofline']['stages_ms']))" || exit 1
done
