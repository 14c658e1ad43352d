"""Tools.toArgb on the GPU against the fp64 oracle for EVERY (Y, Cb, Cr) triple (2^24 of them): the kernels use an
integer form of the reference's double-precision expressions (webp_amd/csrc/kernels/wp_convert.h), which is only
acceptable because it is bit-identical everywhere, including the exact .5 ties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _all_triples_image():
    # 2048 x 8192 luma, 1024 x 4096 chroma: Cb = cx % 256, Cr = cy % 256; the 64 chroma cells that share a (Cb, Cr)
    # pair carry the 256 luma values in their 2x2 luma samples.
    w, h = 2048, 8192
    cx = np.arange(w // 2)[None, :]
    cy = np.arange(h // 2)[:, None]
    u = np.broadcast_to((cx % 256).astype(np.uint8), (h // 2, w // 2)).copy()
    v = np.broadcast_to((cy % 256).astype(np.uint8), (h // 2, w // 2)).copy()
    cell = (cx // 256) + 4 * (cy // 256)                      # 0..63
    px = np.arange(w)[None, :]
    py = np.arange(h)[:, None]
    y = (4 * cell[py // 2, px // 2] + 2 * (py & 1) + (px & 1)).astype(np.uint8)
    seen = np.zeros((256, 256, 256), bool)
    seen[y, u[py // 2, px // 2], v[py // 2, px // 2]] = True
    assert seen.all()
    return w, h, y, u, v


def test_toargb_every_triple_matches_fp64_oracle():
    from oracle import oracle
    from webp_amd import Tools
    from webp_amd.decoder import FrameHeader, Image, Yuv420
    w, h, y, u, v = _all_triples_image()
    exp = oracle.yuv420_to_argb(y.reshape(-1), w, u.reshape(-1), v.reshape(-1), w // 2, None, w, h)
    img = Image(FrameHeader(width=w, height=h), Yuv420(y.reshape(-1), w, u.reshape(-1), v.reshape(-1), w // 2))
    got = Tools.toArgb(img).data.pix
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, "first mismatch at byte %d: got %d want %d" % (bad[0], got[bad[0]], exp[bad[0]])
