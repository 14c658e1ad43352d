"""Parity soak on the GPU box: random packed macroblock buffers (tests/test_gpu_packed_random.make_frame) in random
batch shapes, all filter types, against the oracle, for a time budget.  Not a test (the committed tests use fixed
seeds); run it after kernel changes:   python tools/soak.py [seconds] [first_seed]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_packed_random import make_frame
from oracle import oracle
from webp_amd._lib import Vp8Frame, check, lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
t0 = time.time(); batches = frames_done = mbs = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 40))
    big = rng.random() < 0.2
    sizes = [(int(rng.integers(1, 130 if big else 24)), int(rng.integers(1, 70 if big else 24))) for _ in range(n)]
    ftypes = [int(rng.integers(0, 3)) for _ in range(n)]
    extreme = bool(rng.integers(0, 2))
    frames = [make_frame(rng, w, h, extreme) for (w, h) in sizes]
    desc = (Vp8Frame * n)()
    mb_base = cbase = yo = co = ao = 0
    for i, ((w, h), (mb, coeff, filt)) in enumerate(zip(sizes, frames)):
        d = desc[i]
        d.width, d.height, d.mbw, d.mbh = 16 * w - int(rng.integers(0, 16)), 16 * h - int(rng.integers(0, 16)), w, h
        d.filter_type = ftypes[i]
        d.mb_base, d.coeff_base = mb_base, cbase
        d.y_off, d.u_off, d.v_off, d.argb_off = yo, co, co, ao
        mb_base += w * h; cbase += coeff.size // 16
        yo += 256 * w * h; yo += (-yo) % 256
        co += 64 * w * h; co += (-co) % 256
        ao += 4 * d.width * d.height; ao += (-ao) % 256
    mb_all = np.concatenate([f[0] for f in frames]); co_all = np.concatenate([f[1] for f in frames]); fl_all = np.concatenate([f[2] for f in frames])
    y = np.zeros(yo, np.uint8); u = np.zeros(co, np.uint8); v = np.zeros(co, np.uint8); argb = np.zeros(ao, np.uint8)
    check(lib().wp_vp8_decode_to_argb(n, desc, p(mb_all), mb_base, p(fl_all), p(co_all), cbase, p(y), p(u), p(v), None, p(argb)))
    for i, ((w, h), (mb, coeff, filt)) in enumerate(zip(sizes, frames)):
        exp = oracle.vp8_from_packed(w, h, mb, coeff, filt, ftypes[i])
        d = desc[i]; nmb = w * h
        for name, got, off, sz in (("y", y, d.y_off, 256 * nmb), ("u", u, d.u_off, 64 * nmb), ("v", v, d.v_off, 64 * nmb)):
            if not np.array_equal(got[off:off + sz], exp[name]):
                print("MISMATCH seed %d frame %d (%dx%d MBs, filter %d, extreme %s) plane %s" % (seed, i, w, h, ftypes[i], extreme, name), flush=True)
                sys.exit(1)
        want = oracle.yuv420_to_argb(exp["y"], 16 * w, exp["u"], exp["v"], 8 * w, None, d.width, d.height)
        if not np.array_equal(argb[d.argb_off:d.argb_off + 4 * d.width * d.height], want):
            print("MISMATCH seed %d frame %d ARGB" % (seed, i), flush=True); sys.exit(1)
        frames_done += 1; mbs += nmb
    batches += 1; seed += 1
    if batches % 20 == 0:
        print("%d batches, %d frames, %d macroblocks, %.0f s: all equal" % (batches, frames_done, mbs, time.time() - t0), flush=True)
print("soak ok: %d batches, %d frames, %d macroblocks bit-exact against the oracle (seeds up to %d)" % (batches, frames_done, mbs, seed - 1))
