"""Parity soak of unfilterAlpha (all four filters, raw and RGBA sources, random sizes incl. many 64-row bands) on the
GPU box against the oracle, for a time budget.   python tools/soak_alpha.py [seconds] [first_seed]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from webp_amd._lib import AlphaFrame, lib, check

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
t0 = time.time(); calls = planes = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 9)); rgba = bool(rng.integers(0, 2))
    desc = (AlphaFrame * n)(); srcs, want, so, do = [], [], 0, 0
    for i in range(n):
        tall = rng.random() < 0.3
        w, h, f = int(rng.integers(1, 330)), int(rng.integers(1, 800 if tall else 140)), int(rng.integers(0, 4))
        a = rng.integers(0, 256, w * h, dtype=np.uint8)
        if f == 3 and rng.random() < 0.7: a = (a // 8).astype(np.uint8)
        want.append(oracle.unfilter_alpha(a, w, f))
        if rgba:
            s = rng.integers(0, 256, 4 * w * h, dtype=np.uint8); s[1::4] = a
        else:
            s = a
        d = desc[i]
        d.width, d.height, d.filter, d.src_is_vp8l_rgba, d.src_off, d.dst_off = w, h, f, int(rgba), so, do
        pad = (-s.size) % 16
        srcs.append(np.concatenate([s, np.zeros(pad, np.uint8)])); so += s.size + pad
        do += w * h + int(rng.integers(0, 4)) * (rng.random() < 0.3)        # sometimes a destination that is not dword-aligned
    src = np.concatenate(srcs); out = np.zeros(do + 16, np.uint8)
    check(lib().wp_alpha_unfilter(n, desc, vp(src), src.size, vp(out), out.size))
    for i in range(n):
        d = desc[i]
        if not np.array_equal(out[d.dst_off:d.dst_off + d.width * d.height], want[i]):
            print("MISMATCH seed %d plane %d (%dx%d filter %d rgba %s)" % (seed, i, d.width, d.height, d.filter, rgba), flush=True); sys.exit(1)
        planes += 1
    calls += 1; seed += 1
print("soak ok: %d calls, %d planes bit-exact against the oracle (seeds up to %d)" % (calls, planes, seed - 1))
