"""Parity soak of the VP8L inverse chain on the GPU box: random sizes, transform orders (with and without the fused
patterns), predictor / cross-colour tile sizes, byte orders, several frames per call -- against the oracle, for a time
budget.   python tools/soak_vp8l.py [seconds] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_vp8l import _run_chain, _tiles
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ORDERS = [(0,), (1, 0), (0, 1), (2, 0), (0, 2), (2, 0, 1), (2, 1, 0), (1, 2, 0), (0, 1, 2), (1,), (2,), ()]
t0 = time.time(); calls = frames = px = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    swizzle = int(rng.integers(0, 3))
    cases, want = [], []
    for _ in range(int(rng.integers(1, 7))):
        big = rng.random() < 0.15
        w, h = int(rng.integers(1, 700 if big else 90)), int(rng.integers(1, 400 if big else 200))
        order = ORDERS[int(rng.integers(0, len(ORDERS)))]
        pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
        bits = {0: int(rng.integers(2, 7)), 1: int(rng.integers(2, 7)), 2: 0}
        data = {0: _tiles(rng, w, h, bits[0], list(range(16))), 1: _tiles(rng, w, h, bits[1]), 2: np.zeros(0, np.uint8)}
        cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(t, bits[t], w, data[t]) for t in order]))
        e = pix
        for t in reversed(order):
            if t == 0: e = oracle.inverse_predictor(w, bits[0], data[0], e, h)
            elif t == 1: e = oracle.inverse_cross_color(w, bits[1], data[1], e, h)
            else: e = oracle.inverse_subtract_green(e)
        if swizzle == 1: e = np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1)
        elif swizzle == 2: e = e[1::4]
        want.append(e)
    got = _run_chain(cases, swizzle=swizzle)
    for i, (c, g, e) in enumerate(zip(cases, got, want)):
        if not np.array_equal(g[:e.size], e):
            print("MISMATCH seed %d frame %d (%dx%d, order %s, swizzle %d)" % (seed, i, c["width"], c["height"], [t[0] for t in c["transforms"]], swizzle), flush=True)
            sys.exit(1)
        frames += 1; px += c["width"] * c["height"]
    calls += 1; seed += 1
    if calls % 50 == 0:
        print("%d calls, %d frames, %d pixels, %.0f s: all equal" % (calls, frames, px, time.time() - t0), flush=True)
print("soak ok: %d calls, %d frames, %d pixels bit-exact against the oracle (seeds up to %d)" % (calls, frames, px, seed - 1))
