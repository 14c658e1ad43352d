#!/bin/bash
# Builds tuning variants of the library: tools/build_variants.sh name "-DFLAGS" [name "-DFLAGS" ...]
# -> webp_amd/variants/libwebp_amd_<name>.so (git-ignored; travels to the GPU box), for WEBP_AMD_LIB=...
set -e
cd "$(dirname "$0")/../webp_amd/csrc"
mkdir -p ../variants
while [ $# -ge 2 ]; do
  make -s -j8 OUT=../variants/libwebp_amd_$1.so BUILD=/tmp/wpbuild_$1 VARIANT="$2" 2>&1 | grep -E "error|Error" || true
  echo "built $1 ($2)"
  shift 2
done
