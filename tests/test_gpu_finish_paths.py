"""GPU parity of the finishing pass's two code paths (vp8_finish_chunk.h): chunks that lie wholly inside the image take a
predicate-free path with a pixel-major conversion, everything else -- frame edges, widths that are not a multiple of 4,
frames with alpha, and any wavefront that meets the chroma pair whose green term needs the Y-dependent tie correction
(wp_convert.h) -- the general one.  The frames here are built so that each case occurs where it is known to: shapes with
interior chunks, ragged right / bottom edges, and frames whose chroma is the tie pair (Cb, Cr) = (78, 178) in every sample
while luma sweeps the whole range.  Checked against the oracle's packed-buffer hook + its fp64 toArgb; through the C ABI."""
import ctypes
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (mbw, mbh, visible width, visible height)
SHAPES = [(16, 4, 256, 64),      # two full chunks, tile rows 1..3 interior
          (17, 5, 270, 77),      # width % 4 != 0: never the interior path
          (24, 3, 384, 48),      # three full chunks
          (9, 6, 144, 96),       # chunk 0 interior, chunk 1 ragged
          (8, 2, 128, 32),       # exactly one chunk; only tile row 1 is interior
          (40, 5, 636, 80)]      # width % 4 == 0 but not a multiple of 128


def tie_frame(rng, mbw, mbh):
    """Every macroblock i16 / DC-predicted; chroma 78 / 178 everywhere (set once through the first macroblock's chroma DCs and
    carried by DC prediction with no residual); luma from random WHT coefficients and a few random AC blocks."""
    n = mbw * mbh
    mb = np.zeros((n, 4), np.uint32)
    blocks = []
    nblk = 0
    for i in range(n):
        present = np.zeros(24, bool)
        present[:16] = rng.random(16) < 0.2
        if i == 0:
            present[16:24] = True
        w2 = sum(1 << b for b in range(24) if present[b]) | (1 << 24) | (1 << 25)      # WHT + i16, predC8 = DC (0)
        mb[i, 0] = 0                                                                   # predY16 = DC
        mb[i, 2] = w2
        mb[i, 3] = nblk
        wht = np.rint(rng.laplace(0, 300, 16)).astype(np.int64)
        wht[0] = int(rng.integers(-6000, 6000))
        blocks.append(np.clip(wht, -32768, 32767).astype(np.int16))
        for b in range(24):
            if not present[b]:
                continue
            c = np.zeros(16, np.int64)
            if b < 16:
                c[1:] = np.rint(rng.laplace(0, 30, 15))
                c[rng.random(16) < 0.7] = 0
                c[0] = 0
            else:
                c[0] = -400 if b < 20 else 400          # 128 + ((-400 + 4) >> 3) = 78, 128 + ((400 + 4) >> 3) = 178
            blocks.append(c.astype(np.int16))
        nblk += 1 + int(present.sum())
    level = rng.integers(0, 40, n)
    filt = (level | (rng.integers(0, 10, n) << 8) | (rng.integers(0, 3, n) << 16) | (rng.integers(0, 2, n) << 24)).astype(np.uint32)
    return mb.reshape(-1), np.concatenate(blocks), filt


@pytest.mark.parametrize("kind", ["random", "tie"])
@pytest.mark.parametrize("filter_type", [0, 2])
def test_interior_and_general_paths(kind, filter_type):
    from packed_util import RawBatch
    from test_gpu_packed_random import make_frame
    from webp_amd._lib import check, lib
    rng = np.random.default_rng(77 + filter_type + (kind == "tie"))
    shapes = [(w, h) for (w, h, _, _) in SHAPES]
    frames = [tie_frame(rng, w, h) if kind == "tie" else make_frame(rng, w, h, False) for (w, h) in shapes]
    b = RawBatch(shapes, frames, [filter_type] * len(frames), visible=[(vw, vh) for (_, _, vw, vh) in SHAPES])
    y = np.zeros(b.y_bytes, np.uint8); u = np.zeros(b.u_bytes, np.uint8); v = np.zeros(b.v_bytes, np.uint8)
    argb = np.zeros(b.argb_bytes, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8_decode_to_argb(b.n, b.desc, p(b.mb), b.n_mb, p(b.filt), p(b.coeff), b.n_coeff_blocks, p(y), p(u), p(v), None, p(argb)))
    for i, (w, h) in enumerate(shapes):
        exp = b.expected(i)
        d = b.desc[i]
        if kind == "tie":
            assert set(np.unique(exp["u"])) == {78} and set(np.unique(exp["v"])) == {178}, "the frame is not the tie pair throughout"
            lum = np.unique(exp["y"])
            assert lum.min() < 111 and lum.max() > 146 and np.count_nonzero((lum >= 111) & (lum <= 146)) > 20
        assert np.array_equal(y[d.y_off:d.y_off + 256 * w * h], exp["y"]), "frame %d Y" % i
        assert np.array_equal(u[d.u_off:d.u_off + 64 * w * h], exp["u"]) and np.array_equal(v[d.v_off:d.v_off + 64 * w * h], exp["v"]), "frame %d chroma" % i
        got = argb[d.argb_off:d.argb_off + 4 * d.width * d.height]
        bad = np.flatnonzero(got != exp["argb"])
        assert bad.size == 0, "frame %d (%dx%d): %d ARGB bytes differ, first at pixel (%d, %d) channel %d: got %d want %d" % (
            i, d.width, d.height, bad.size, (bad[0] // 4) % d.width, (bad[0] // 4) // d.width, bad[0] % 4, got[bad[0]], exp["argb"][bad[0]])


@pytest.mark.parametrize("misalign", [0, 1])
def test_alpha_planes_through_both_paths(misalign):
    """Frames with an alpha plane: interior chunks take the alpha bytes as one dword per four pixels and patch them into the
    converted pixels; ragged chunks, widths that are not a multiple of 4 and planes that do not start on a 4-byte boundary
    (misalign) take the general path with its byte loads."""
    from packed_util import RawBatch
    from test_gpu_packed_random import make_frame
    from oracle import oracle
    from webp_amd._lib import check, lib
    rng = np.random.default_rng(4321 + misalign)
    shapes = [(w, h) for (w, h, _, _) in SHAPES]
    frames = [make_frame(rng, w, h, False) for (w, h) in shapes]
    b = RawBatch(shapes, frames, [2] * len(frames), visible=[(vw, vh) for (_, _, vw, vh) in SHAPES])
    ao = misalign
    alphas = []
    for i in range(b.n):
        d = b.desc[i]
        d.has_alpha, d.alpha_off = 1, ao
        a = rng.integers(0, 256, d.width * d.height, dtype=np.uint8)
        a[rng.random(a.size) < 0.3] = 255
        alphas.append(a)
        ao += (a.size + 255) // 256 * 256
    alpha = np.zeros(ao + 256, np.uint8)
    for i in range(b.n):
        alpha[b.desc[i].alpha_off:b.desc[i].alpha_off + alphas[i].size] = alphas[i]
    y = np.zeros(b.y_bytes, np.uint8); u = np.zeros(b.u_bytes, np.uint8); v = np.zeros(b.v_bytes, np.uint8)
    argb = np.zeros(b.argb_bytes, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8_decode_to_argb(b.n, b.desc, p(b.mb), b.n_mb, p(b.filt), p(b.coeff), b.n_coeff_blocks, p(y), p(u), p(v), p(alpha), p(argb)))
    for i, (w, h) in enumerate(shapes):
        exp = b.expected(i)
        d = b.desc[i]
        want = oracle.yuv420_to_argb(exp["y"], 16 * w, exp["u"], exp["v"], 8 * w, alphas[i], d.width, d.height)
        got = argb[d.argb_off:d.argb_off + 4 * d.width * d.height]
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, "frame %d (%dx%d): %d ARGB bytes differ, first at pixel (%d, %d) channel %d: got %d want %d" % (
            i, d.width, d.height, bad.size, (bad[0] // 4) % d.width, (bad[0] // 4) // d.width, bad[0] % 4, got[bad[0]], want[bad[0]])
