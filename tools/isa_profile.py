#!/usr/bin/env python3
"""Static VALU profile of a kernel's ISA by source line.

usage: tools/isa_profile.py <file.hip> [kernel-substring] [-D...]
Compiles the file for gfx950 with line tables (-gline-tables-only -S), walks the assembly of the kernel whose symbol
contains the substring, and prints VALU instruction counts per source file and line range, split into the instructions that
issue in 2 cycles per wavefront and those that take 4 (tools/ubench/valu_rate*.hip; profiles/r01_ubench_valu_rate.txt).
Static counts: code under wave-uniform branches (the i4 mini-steps, the full chroma transform) is counted once.
"""
import collections
import re
import subprocess
import sys
import tempfile

CHEAP = {  # 2 cycles per wave64 instruction on gfx950 (measured)
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_or_b32", "v_xor_b32", "v_and_b32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_mov_b32", "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_max_i16", "v_min_i16", "v_max_u16", "v_min_u16",
    "v_lshlrev_b16", "v_lshrrev_b16", "v_ashrrev_i16", "v_mul_lo_u16", "v_add_f32", "v_sub_f32", "v_subrev_f32",
    "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_cndmask_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32",
    "v_accvgpr_write_b32", "v_accvgpr_read_b32",
}


def main():
    src = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    defs = [a for a in sys.argv[2:] if a.startswith("-")]
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-gline-tables-only", "-S", "--cuda-device-only", src, "-o", tmp.name] + defs
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(tmp.name).read().splitlines()
    files = {}
    cur = ("?", 0)
    in_kernel = False
    per_line = collections.Counter()
    per_line_cost = collections.Counter()
    ops = collections.Counter()
    other = collections.Counter()
    for ln in text:
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', s)
        if m:
            files[int(m.group(1))] = m.group(3).split("/")[-1]
            continue
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"', s)
        if m:
            files[int(m.group(1))] = m.group(2).split("/")[-1]
            continue
        if re.match(r"^[A-Za-z_][\w.$]*:", s) and not s.startswith(".L"):
            name = s.split(":")[0]
            in_kernel = (sub in name) and not name.startswith("__")
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            in_kernel = False
        if not in_kernel:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith(".") or s.startswith(";"):
            continue
        op = s.split()[0]
        base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
        if op.startswith("v_"):
            cost = 1 if (base in CHEAP and not op.endswith(("_sdwa", "_dpp"))) else 2
            per_line[cur] += 1
            per_line_cost[cur] += cost
            ops[base] += 1
        else:
            kind = "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_", "flat_", "scratch_")) else "other"
            other[(cur[0], kind)] += 1
            other[("*", kind)] += 1
    by_file = collections.Counter()
    by_file_cost = collections.Counter()
    for (f, l), n in per_line.items():
        by_file[f] += n
        by_file_cost[f] += per_line_cost[(f, l)]
    print("== VALU per file (static): instructions, half-cost units (2-cycle op = 1, 4-cycle op = 2)")
    for f, n in by_file.most_common():
        print(f"  {f:28s} {n:6d} {by_file_cost[f]:7d}")
    print(f"  {'total':28s} {sum(by_file.values()):6d} {sum(by_file_cost.values()):7d}")
    print("== other instruction kinds:", {k[1]: v for k, v in other.items() if k[0] == "*"})
    topn = int(__import__("os").environ.get("ISA_TOP", "60"))
    print("== VALU per line (top %d)" % topn)
    for (f, l), n in sorted(per_line.items(), key=lambda kv: -kv[1])[:topn]:
        print(f"  {f}:{l:<5d} {n:5d} {per_line_cost[(f, l)]:6d}")
    print("== opcodes (top 40)")
    for o, n in ops.most_common(40):
        print(f"  {o:24s} {n:5d} {'cheap' if o in CHEAP else ''}")


if __name__ == "__main__":
    main()
