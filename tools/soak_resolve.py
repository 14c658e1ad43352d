"""GPU parity soak for wp_vp8l_resolve (decodePix's pixel loop): random valid token streams of all three styles
(tests/test_gpu_vp8l_resolve._random_stream) in random image sizes and cache sizes against the oracle's loop, for a time
budget.   python tools/soak_resolve.py [seconds] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_vp8l_resolve import _random_stream, _resolve
from oracle import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
n_img = n_px = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(int(rng.integers(1, 60))):
        style = int(rng.integers(0, 3))
        if style == 1:
            w, h = int(rng.integers(200, 3500)), int(rng.integers(8, 60))
        else:
            w, h = int(rng.integers(1, 400)), int(rng.integers(1, 120))
        cc = int(rng.integers(0, 12))
        cases.append((w, h, cc, _random_stream(rng, w, h, cc, style)))
    got = _resolve(cases)
    for i, (w, h, cc, t) in enumerate(cases):
        want = oracle.resolve_tokens(w, h, cc, t)
        if not np.array_equal(got[i], want):
            print("MISMATCH seed %d stream %d (%dx%d, cache bits %d)" % (seed, i, w, h, cc)); sys.exit(1)
        n_img += 1; n_px += w * h
    seed += 1
    if seed % 20 == 0:
        print("seed %d: %d images, %.1f Mpix, all equal" % (seed, n_img, n_px / 1e6), flush=True)
print("soak_resolve: %d images / %.1f Mpix in %.0f s (seeds up to %d), all equal" % (n_img, n_px / 1e6, time.time() - t0, seed - 1))
