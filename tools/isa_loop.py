#!/usr/bin/env python3
"""VALU instructions of vp8_mb_kernel<8> inside the step loop of mb_band (static count by call site in vp8_mb_band.h; code
under wave-uniform branches -- i4 mini-steps, full chroma transform, simple filter -- is counted once).
usage: tools/isa_loop.py [-D...]     prints: loop VALU instructions, half-cost units, and the same before the loop"""
import collections, re, subprocess, sys, tempfile
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_profile import CHEAP

src = "webp_amd/csrc/kernels/vp8_lossy.hip"
text = open("webp_amd/csrc/kernels/vp8_mb_band.h").read().split("\n")
loop_line = next(i + 1 for i, l in enumerate(text) if "for (int s = 0; s < n_steps; ++s)" in l)
with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
                           "--cuda-device-only", "-gline-tables-only", "-S", "-o", tmp.name, src] + [a for a in sys.argv[1:] if a.startswith("-")],
                          stderr=subprocess.DEVNULL)
    asm = open(tmp.name).read().split("\n")
cur, inside = None, False
n = collections.Counter(); u = collections.Counter()
for ln in asm:
    if ln.startswith("_ZN3wpk13vp8_mb_kernelILi8"):
        inside = True
    if inside and "s_endpgm" in ln:
        break
    if not inside:
        continue
    if ln.lstrip().startswith(".loc"):
        mb = [int(l) for f, l in re.findall(r"(\w+\.h|\w+\.hip):(\d+)", ln) if f == "vp8_mb_band.h"]
        cur = mb[-1] if mb else cur
        continue
    m = re.match(r"\s+(v_\w+)", ln)
    if m:
        op = re.sub(r"_e32$|_e64$", "", m.group(1))
        key = "loop" if cur is not None and cur >= loop_line else "before"
        n[key] += 1
        u[key] += 1 if op in CHEAP else 2
print("step loop starts at vp8_mb_band.h:%d" % loop_line)
for k in ("loop", "before"):
    print("%-7s %5d VALU instructions, %5d half-cost units" % (k, n[k], u[k]))
