"""BASELINE.json configs[3] as a test: ONE animated file of 300 ANMF frames of 1280x720, lossy with alpha, decoded through
the public surface (WebPDecoder.decode -> one device batch; per-frame conversion as test/Visual.hx does) against the oracle.
The frames' ALPH chunks cycle through every combination the reference reads (WebPDecoder.hx:78-83, 311-411): VP8L-coded and
raw planes x prediction filters none / horizontal / vertical / gradient (libwebp itself only writes VP8L-coded planes with
filter 0 or 1 here; the other headers are re-written by tools/make_anim.py, so the planes they yield are not an encoder's --
parity unpinned by a golden for filters 1-3, as SURVEY 8c says: oracle-vs-device only)."""
import glob
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_300_frame_animation_with_every_alpha_form():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_anim import alph_with_filter, build_anim, raw_alph
    from oracle import oracle
    from webp_amd import Anim, Tools, WebPDecoder
    stills = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy720_alpha_*.webp")))]
    assert stills, "bench_data/lossy720_alpha_*.webp missing"
    rng = np.random.default_rng(300)
    w, h = 1280, 720
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((xx // 5 + yy // 3) & 0x0f).astype(np.uint8).tobytes()       # small deltas: what a filtered plane looks like
    noisy = rng.integers(0, 256, w * h, dtype=np.uint8).tobytes()

    def alph_for(k, alph, fw, fh):
        assert (fw, fh) == (w, h) and (alph[0] & 3) == 1                   # the stills carry VP8L-coded planes
        form = k % 8
        if form < 4:
            return alph_with_filter(alph, form)
        return raw_alph(noisy if k % 16 >= 8 else smooth, form - 4)

    n = 300
    data = build_anim(stills, n, alph_for=alph_for)
    img = WebPDecoder.decode(data)
    assert isinstance(img.data, Anim) and len(img.data.frames) == n
    got = Tools.framesToArgb(img)
    want = oracle.decode(data)["frames"]
    assert len(want) == n
    check = sorted(set(list(range(16)) + [149, 150, 151, 255, 256, 257] + list(range(n - 8, n))))      # every form, both stills, both ends, the CU-count boundary
    for k in check:
        f = img.data.frames[k]
        assert np.array_equal(f.data.a, want[k]["a"]), "alpha plane of frame %d (form %d)" % (k, k % 8)
        assert np.array_equal(f.data.y, want[k]["y"]) and np.array_equal(f.data.u, want[k]["u"]) and np.array_equal(f.data.v, want[k]["v"]), "planes of frame %d" % k
        assert np.array_equal(got[k], want[k]["to_argb"]), "A,R,G,B of frame %d (form %d)" % (k, k % 8)
