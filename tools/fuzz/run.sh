#!/bin/bash
# Host-parser fuzz under AddressSanitizer + UBSan (host code only): mutates the fixtures, expects clean rejections.
set -e
cd "$(dirname "$0")/../../webp_amd/csrc"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I. -I../../include \
    ../../tools/fuzz/parse_fuzz.cpp host/vp8_parse.cpp host/vp8l_parse.cpp host/webp_parse.cpp -o /tmp/wp_parse_fuzz
cd ../..
/tmp/wp_parse_fuzz tests/golden/inputs/*.webp bench_data/lossy_*.webp bench_data/lossy720_alpha_0.webp
