"""Canvas compositing.  The reference does none (Image.hx:21-34 hands AnimFrameHeader to the caller), so
the pin is libwebp's WebPAnimDecoder: tests/golden/anim/*.npz hold the canvases it composes (through PIL, generated in the
build container by tests/golden/make_anim_golden.py) from the lossless animations beside them.  CPU: the oracle's
restatement against those canvases.  GPU: wp_anim_compose against the same canvases and, for lossy frames with alpha and
offsets no encoder produced, against the oracle."""
import ctypes
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "anim", "*.webp")))


def _golden_argb(path):
    rgba = np.load(path[:-5] + ".npz")["rgba"]
    return np.ascontiguousarray(rgba[..., [3, 0, 1, 2]])


def _oracle_frames(data):
    from oracle import oracle
    img = oracle.decode(data)
    assert img["kind"] == "anim"
    return img, [(f["x"], f["y_off"], f["width"], f["height"], f["blend"], f["dispose"], f["to_argb"]) for f in img["frames"]]


def test_goldens_exist():
    assert len(FILES) == 16


@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_oracle_compositing_matches_libwebp(path):
    from oracle import oracle
    img, frames = _oracle_frames(open(path, "rb").read())
    got = oracle.anim_compose(frames, img["width"], img["height"])
    want = _golden_argb(path)
    assert got.shape == want.shape
    for k in range(got.shape[0]):
        assert np.array_equal(got[k], want[k]), "canvas %d of %s" % (k, os.path.basename(path))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FILES, ids=os.path.basename)
def test_gpu_compositing_matches_libwebp(path):
    from webp_amd import Tools, WebPDecoder
    got = Tools.composeAnim(WebPDecoder.decode(open(path, "rb").read()))
    want = _golden_argb(path)
    assert len(got) == want.shape[0]
    for k in range(len(got)):
        assert np.array_equal(got[k].reshape(want[k].shape), want[k]), "canvas %d of %s" % (k, os.path.basename(path))


@pytest.mark.gpu
def test_gpu_compositing_random_rectangles_against_the_oracle():
    """Frames with arbitrary offsets (clipped at the canvas edge), sizes, flags and alpha, which no encoder produced."""
    from oracle import oracle
    from webp_amd._lib import AnimFrame, check, lib
    rng = np.random.default_rng(99)
    for trial in range(12):
        cw, ch = int(rng.integers(1, 90)), int(rng.integers(1, 70))
        n = int(rng.integers(1, 9))
        desc = (AnimFrame * n)()
        frames, bufs, off = [], [], 0
        for k in range(n):
            full = rng.random() < 0.25
            w, h = (cw, ch) if full else (int(rng.integers(1, cw + 8)), int(rng.integers(1, ch + 8)))
            x, y = (0, 0) if full else (int(rng.integers(0, cw // 2 + 1)), int(rng.integers(0, ch // 2 + 1)))
            pix = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            pix[..., 0] = rng.choice(np.array([0, 255, 1, 128, 254], np.uint8), (h, w), p=[0.25, 0.25, 0.1, 0.3, 0.1])
            blend, dispose = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            d = desc[k]
            d.x, d.y, d.width, d.height, d.blend, d.dispose, d.argb_off = x, y, w, h, blend, dispose, off
            frames.append((x, y, w, h, blend, dispose, pix.reshape(-1)))
            bufs.append(pix.reshape(-1))
            off += pix.size
        slab = np.concatenate(bufs)
        out = np.zeros(n * ch * cw * 4, np.uint8)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        check(lib().wp_anim_compose(n, desc, cw, ch, vp(slab), slab.size, vp(out), out.size))
        want = oracle.anim_compose(frames, cw, ch)
        assert np.array_equal(out.reshape(want.shape), want), "trial %d" % trial


@pytest.mark.gpu
def test_gpu_compositing_of_lossy_frames_with_alpha():
    """cfg 4's animated file (lossy frames + VP8L-coded ALPH, built by tools/make_anim.py): decode on the device, compose on
    the device, against the oracle's decode + compositing."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_anim import build_anim
    from oracle import oracle
    from webp_amd import Tools, WebPDecoder
    stills = sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy720_alpha_*.webp")))
    if not stills:
        pytest.skip("bench_data/lossy720_alpha_*.webp missing")
    data = build_anim([open(p, "rb").read() for p in stills], 4)
    got = Tools.composeAnim(WebPDecoder.decode(data))
    img, frames = _oracle_frames(data)
    want = oracle.anim_compose(frames, img["width"], img["height"])
    assert len(got) == want.shape[0]
    for k in range(len(got)):
        assert np.array_equal(got[k].reshape(want[k].shape), want[k]), "canvas %d" % k
