#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the lossy kernels for one library build: tools/traffic_ab.sh <tag> [lib.so]
tag=$1; lib=${2:-}
[ -n "$lib" ] && export WEBP_AMD_LIB=$PWD/$lib
PMC_LIST=tools/pmc_traffic.list tools/pmc.sh $tag --serial --steps 2 --warmup 1 --no-secondary > gpurun_out/traffic_$tag.log 2>&1 || { tail -5 gpurun_out/traffic_$tag.log; exit 1; }
python3 tools/traffic_json.py gpurun_out/pmc_$tag 1024 gpurun_out/traffic_$tag.json | grep -A4 "vp8_mb_kernel"
