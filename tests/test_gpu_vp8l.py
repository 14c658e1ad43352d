"""GPU parity of the individual VP8L inverse transforms and the alpha un-filters on seeded random
inputs (every predictor mode incl. the 14/15 no-ops, cross-colour sign handling, all pixel-packing
widths, ragged sizes), against the oracle's restatement of Transform.hx / unfilterAlpha."""
import ctypes
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_chain(cases, swizzle=0):
    """cases: list of dict(width, height, coded_width, pix, transforms=[(type,bits,old_width,data)])"""
    from webp_amd._lib import Vp8lFrame, lib, check
    n = len(cases)
    desc = (Vp8lFrame * n)()
    pix, aux = [], []
    po = ao = oo = 0
    al = lambda v, a: (v + a - 1) // a * a
    for i, c in enumerate(cases):
        d = desc[i]
        d.width, d.height, d.coded_width = c["width"], c["height"], c["coded_width"]
        d.n_transforms = len(c["transforms"])
        d.pix_off, d.out_off, d.tmp_off = po, oo, oo
        for k, (tt, bits, ow, data) in enumerate(c["transforms"]):
            t = d.t[k]
            t.type, t.bits, t.old_width, t.data_off, t.data_len = tt, bits, ow, ao, data.size
            padded = np.zeros(al(max(data.size, 1), 16), np.uint8); padded[:data.size] = data
            aux.append(padded); ao += padded.size
        p = np.zeros(al(c["pix"].size, 256), np.uint8); p[:c["pix"].size] = c["pix"]
        pix.append(p); po += p.size
        oo += al(4 * c["width"] * c["height"], 256)
    pix = np.concatenate(pix); aux = np.concatenate(aux) if aux else np.zeros(16, np.uint8)
    out = np.zeros(oo, np.uint8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8l_inverse(n, desc, vp(pix), pix.size, vp(aux), aux.size, vp(out), out.size, swizzle))
    return [out[desc[i].out_off:desc[i].out_off + 4 * cases[i]["width"] * cases[i]["height"]] for i in range(n)]


def _tiles(rng, w, h, bits, modes=None):
    tw, th = (w + (1 << bits) - 1) >> bits, (h + (1 << bits) - 1) >> bits
    t = rng.integers(0, 256, (th * tw, 4), dtype=np.uint8)
    if modes is not None:
        t[:, 1] = (t[:, 1] & 0xF0) | rng.choice(modes, th * tw).astype(np.uint8)
    return t.reshape(-1)


SIZES = [(1, 1), (2, 3), (5, 1), (1, 7), (37, 23), (64, 64), (67, 130), (256, 70), (301, 129)]


@pytest.mark.parametrize("bits", [2, 3, 5])
def test_predictor_all_modes(bits):
    from oracle import oracle
    rng = np.random.default_rng(100 + bits)
    cases, want = [], []
    for (w, h) in SIZES:
        for modes in (list(range(16)), [11, 12, 13], [5, 6, 7, 8, 9, 10], [0, 1, 2, 3, 4, 14, 15]):
            pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
            if rng.integers(0, 2):            # smooth-ish data drives the clamp / select branches harder
                pix = (pix // 32).astype(np.uint8)
            t = _tiles(rng, w, h, bits, modes)
            cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(0, bits, w, t)]))
            want.append(oracle.inverse_predictor(w, bits, t, pix, h))
    got = _run_chain(cases)
    for c, g, e in zip(cases, got, want):
        assert np.array_equal(g, e), (c["width"], c["height"])


@pytest.mark.parametrize("swizzle", [0, 1])
def test_predictor_one_or_two_modes_per_image(swizzle):
    """The predictor's fast paths: every lane of a wavefront with the same mode (each of the 16, the ones that read TR
    included), or select and clamp-add-subtract mixed tile by tile -- widths that end on a full group, in mid-group and
    mid-chunk, more than one band, with and without a cross-colour before the predictor (fused into its loads)."""
    from oracle import oracle
    rng = np.random.default_rng(4100 + swizzle)
    cases, want = [], []
    for (w, h) in [(64, 70), (67, 130), (129, 66), (40, 200), (200, 65)]:
        for modes in [[m] for m in range(16)] + [[11, 12], [10, 11]]:
            bits = int(rng.integers(2, 5))
            pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
            if rng.integers(0, 2):
                pix = (pix // 32).astype(np.uint8)
            t = _tiles(rng, w, h, bits, modes)
            tr = [(0, bits, w, t)]
            e = oracle.inverse_predictor(w, bits, t, pix, h)
            if rng.integers(0, 2):            # [predictor, cross-colour] in the bitstream: cross-colour is undone first
                cb = int(rng.integers(2, 5))
                ct = _tiles(rng, w, h, cb)
                tr = [(0, bits, w, t), (1, cb, w, ct)]
                e = oracle.inverse_predictor(w, bits, t, oracle.inverse_cross_color(w, cb, ct, pix, h), h)
            cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=tr))
            if swizzle == 1:
                e = np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1)
            want.append(e)
    got = _run_chain(cases, swizzle=swizzle)
    for c, g, e in zip(cases, got, want):
        assert np.array_equal(g, e), (c["width"], c["height"], [t[:2] for t in c["transforms"]])


def test_cross_color_and_subtract_green():
    from oracle import oracle
    rng = np.random.default_rng(5)
    cases, want = [], []
    for (w, h) in SIZES:
        for bits in (2, 4):
            pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
            t = _tiles(rng, w, h, bits)
            cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(1, bits, w, t)]))
            want.append(oracle.inverse_cross_color(w, bits, t, pix, h))
        pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
        cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(2, 0, w, np.zeros(0, np.uint8))]))
        want.append(oracle.inverse_subtract_green(pix))
    for g, e in zip(_run_chain(cases), want):
        assert np.array_equal(g, e)


@pytest.mark.parametrize("bits", [0, 1, 2, 3])
def test_color_indexing(bits):
    from oracle import oracle
    rng = np.random.default_rng(9 + bits)
    cases, want = [], []
    for (w, h) in SIZES:
        cw = (w + (1 << bits) - 1) >> bits
        pix = rng.integers(0, 256, 4 * cw * h, dtype=np.uint8)
        pal = np.zeros(4 * (1 << (8 >> bits)), np.uint8)
        pal[:] = rng.integers(0, 256, pal.size, dtype=np.uint8)
        cases.append(dict(width=w, height=h, coded_width=cw, pix=pix, transforms=[(3, bits, w, pal)]))
        want.append(oracle.inverse_color_indexing(w, bits, pal, pix, h))
    for g, e in zip(_run_chain(cases), want):
        assert np.array_equal(g, e)


def test_full_chain_with_swizzle():
    """predictor + cross-colour + subtract-green in libwebp's bitstream order, then RGBA->ARGB."""
    from oracle import oracle
    rng = np.random.default_rng(77)
    w, h = 203, 131
    pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    tp, tc = _tiles(rng, w, h, 3, list(range(14))), _tiles(rng, w, h, 4)
    chain = [(2, 0, w, np.zeros(0, np.uint8)), (0, 3, w, tp), (1, 4, w, tc)]       # bitstream order
    e = oracle.inverse_cross_color(w, 4, tc, pix, h)
    e = oracle.inverse_predictor(w, 3, tp, e, h)
    e = oracle.inverse_subtract_green(e).reshape(-1, 4)
    e = np.ascontiguousarray(e[:, [3, 0, 1, 2]]).reshape(-1)
    g = _run_chain([dict(width=w, height=h, coded_width=w, pix=pix, transforms=chain)], swizzle=1)[0]
    assert np.array_equal(g, e)


@pytest.mark.parametrize("swizzle", [0, 1])
@pytest.mark.parametrize("order", [(1, 0), (0, 2), (2, 1, 0), (1, 2, 0), (0, 1, 2), (2, 0), (0, 1), (2, 0, 1)])
def test_transform_orders_and_fused_passes(order, swizzle):
    """The kernels fold a cross-colour undone right before a predictor, a subtract-green undone right after it and the
    final copy / rotation into the predictor pass (and the final copy into a last colour-indexing pass); every other
    order takes one pass per transform.  Transform lists in orders no encoder emits, 150 rows (three 64-row bands: the
    band hand-over reads raw predictor output back), both byte orders."""
    from oracle import oracle
    rng = np.random.default_rng(100 + sum(order) + 7 * len(order) + swizzle)
    w, h = 77, 150
    pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    data = {0: _tiles(rng, w, h, 3, list(range(14))), 1: _tiles(rng, w, h, 2), 2: np.zeros(0, np.uint8)}
    bits = {0: 3, 1: 2, 2: 0}
    chain = [(t, bits[t], w, data[t]) for t in order]                 # bitstream order
    e = pix
    for t in reversed(order):                                         # undone last-read-first (Vp8LDecoder.hx:395-401)
        if t == 0: e = oracle.inverse_predictor(w, bits[0], data[0], e, h)
        elif t == 1: e = oracle.inverse_cross_color(w, bits[1], data[1], e, h)
        else: e = oracle.inverse_subtract_green(e)
    if swizzle:
        e = np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1)
    g = _run_chain([dict(width=w, height=h, coded_width=w, pix=pix, transforms=chain)], swizzle=swizzle)[0]
    assert np.array_equal(g, e)


@pytest.mark.parametrize("swizzle", [0, 1])
def test_palette_after_predictor(swizzle):
    """Colour indexing without pixel packing (bits 0) as the last transform undone, after a predictor on the index
    image (libwebp's palette + spatial mode): the indexing pass writes the result slab itself."""
    from oracle import oracle
    rng = np.random.default_rng(9 + swizzle)
    w, h = 45, 70
    pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    pal = rng.integers(0, 256, 4 * 256, dtype=np.uint8)
    tp = _tiles(rng, w, h, 2, list(range(14)))
    e = oracle.inverse_color_indexing(w, 0, pal, oracle.inverse_predictor(w, 2, tp, pix, h), h)
    if swizzle:
        e = np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1)
    g = _run_chain([dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(3, 0, w, pal), (0, 2, w, tp)])], swizzle=swizzle)[0]
    assert np.array_equal(g, e)


def test_green_only_output_is_the_green_channel():
    """WP_SWIZZLE_GREEN: the last pass keeps what readAlpha keeps of a VP8L-coded ALPH image (WebPDecoder.hx:330-336).
    Sizes cover every pixel count mod 4 (the kernel packs four pixels per store) and a colour-indexed image, whose last
    pass reads the temporary slab."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    cases, want = [], []
    for (w, h) in [(1, 1), (2, 1), (3, 1), (5, 1), (7, 3), (64, 2), (203, 131)]:
        pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
        t = _tiles(rng, w, h, 3, list(range(14)))
        cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(2, 0, w, np.zeros(0, np.uint8)), (0, 3, w, t)]))
        want.append(oracle.inverse_subtract_green(oracle.inverse_predictor(w, 3, t, pix, h))[1::4])
    w, h, bits = 37, 11, 1
    pw = (w + 1) >> 1
    pix = rng.integers(0, 256, 4 * pw * h, dtype=np.uint8)
    pal = rng.integers(0, 256, 4 * 16, dtype=np.uint8)
    cases.append(dict(width=w, height=h, coded_width=pw, pix=pix, transforms=[(3, bits, w, pal)]))
    want.append(oracle.inverse_color_indexing(w, bits, pal, pix, h)[1::4])
    for (w, h, bits) in ((64, 5, 2), (128, 3, 3), (20, 7, 0), (33, 4, 0)):      # indexing pass writes the green bytes itself: aligned quads, ragged rows, no packing
        pw = (w + (1 << bits) - 1) >> bits
        pix = rng.integers(0, 256, 4 * pw * h, dtype=np.uint8)
        pal = rng.integers(0, 256, 4 * (1 << (8 >> bits)), dtype=np.uint8)
        cases.append(dict(width=w, height=h, coded_width=pw, pix=pix, transforms=[(3, bits, w, pal)]))
        want.append(oracle.inverse_color_indexing(w, bits, pal, pix, h)[1::4])
    got = _run_chain(cases, swizzle=2)
    for c, g, e in zip(cases, got, want):
        n = c["width"] * c["height"]
        assert np.array_equal(g[:n], e), (c["width"], c["height"])


@pytest.mark.parametrize("rgba", [False, True])
def test_alpha_unfilter_modes(rgba):
    from oracle import oracle
    from webp_amd._lib import AlphaFrame, lib, check
    rng = np.random.default_rng(31)
    sizes = [(1, 1), (7, 1), (1, 9), (13, 9), (64, 64), (65, 129), (300, 77), (1280, 70), (40, 600), (131, 321)]      # the last two: more bands than wavefronts
    items = [(w, h, f) for (w, h) in sizes for f in (0, 1, 2, 3)]
    desc = (AlphaFrame * len(items))()
    srcs, so, do, want = [], 0, 0, []
    for i, (w, h, f) in enumerate(items):
        a = rng.integers(0, 256, w * h, dtype=np.uint8)
        if f == 3:
            a = (a // 8).astype(np.uint8)        # small deltas keep the gradient predictor away from pure wrap-around noise
        want.append(oracle.unfilter_alpha(a, w, f))
        if rgba:
            s = rng.integers(0, 256, 4 * w * h, dtype=np.uint8); s[1::4] = a
        else:
            s = a
        d = desc[i]
        d.width, d.height, d.filter, d.src_is_vp8l_rgba, d.src_off, d.dst_off = w, h, f, int(rgba), so, do
        pad = (-s.size) % 16
        srcs.append(np.concatenate([s, np.zeros(pad, np.uint8)])); so += s.size + pad
        do += (w * h + 15) // 16 * 16
    src = np.concatenate(srcs); out = np.zeros(do, np.uint8)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_alpha_unfilter(len(items), desc, vp(src), src.size, vp(out), out.size))
    for i, (w, h, f) in enumerate(items):
        got = out[desc[i].dst_off:desc[i].dst_off + w * h]
        assert np.array_equal(got, want[i]), (w, h, f)


@pytest.mark.parametrize("shape", [(16384, 3), (3, 16384), (16384, 70)])
def test_predictor_maximum_dimensions(shape):
    """VP8L's largest width / height (14 bits + 1): the predictor's progress word holds a pixel count below 2^15, and a
    16384-row image crosses 256 row bands."""
    from oracle import oracle
    w, h = shape
    rng = np.random.default_rng(w + 7 * h)
    bits = 4
    pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    t = _tiles(rng, w, h, bits, list(range(14)))
    got = _run_chain([dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(0, bits, w, t)])])[0]
    assert np.array_equal(got, oracle.inverse_predictor(w, bits, t, pix, h))


def test_inconsistent_transform_lists_are_refused():
    """Every buffer extent the kernels use follows from the transform list, so the entry point checks it before any
    launch: widths must chain from `width` to `coded_width`, tile / palette data must be long enough."""
    from webp_amd._lib import WebPError
    rng = np.random.default_rng(9)
    w, h, bits = 37, 23, 3
    pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
    t = _tiles(rng, w, h, bits)
    ok = dict(width=w, height=h, coded_width=w, pix=pix, transforms=[(0, bits, w, t)])
    assert _run_chain([ok])[0].size == 4 * w * h
    for bad, msg in ((dict(ok, transforms=[(0, bits, w, t[:-4])]), "transform data too short"),
                     (dict(ok, transforms=[(0, bits, w + 1, t)]), "bad transform"),
                     (dict(ok, coded_width=w - 1), "coded_width"),
                     (dict(ok, transforms=[(3, 1, w, np.zeros(60, np.uint8))], coded_width=(w + 1) // 2,
                           pix=np.zeros(4 * ((w + 1) // 2) * h, np.uint8)), "transform data too short")):
        with pytest.raises(WebPError, match=msg):
            _run_chain([bad])
    # dimensions beyond what a VP8L header can hold (14 bits + 1) are refused before anything is sized from them
    from webp_amd._lib import Vp8lFrame, lib, check
    desc = (Vp8lFrame * 1)()
    desc[0].width, desc[0].height, desc[0].coded_width = 16385, 1, 16385
    buf = np.zeros(4 * 16385 + 64, np.uint8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    with pytest.raises(WebPError, match="bad wp_vp8l_frame"):
        check(lib().wp_vp8l_inverse(1, desc, vp(buf), buf.size, vp(buf), 16, vp(buf.copy()), buf.size, 0))


@pytest.mark.parametrize("shape", [(32769, 130), (40001, 70)])
def test_alpha_unfilter_planes_wider_than_32767(shape):
    """A raw ALPH plane takes its size from the 24-bit VP8X canvas (WebPDecoder.hx:133-148): widths beyond 15 bits, rows not
    dword-aligned (the one-pixel-skew row wavefront), more than one 64-row band."""
    from oracle import oracle
    from webp_amd._lib import AlphaFrame, lib, check
    w, h = shape
    rng = np.random.default_rng(w)
    for f in (1, 3):
        a = rng.integers(0, 256, w * h, dtype=np.uint8)
        if f == 3:
            a = (a // 8).astype(np.uint8)
        desc = (AlphaFrame * 1)()
        desc[0].width, desc[0].height, desc[0].filter = w, h, f
        out = np.zeros(w * h, np.uint8)
        vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        check(lib().wp_alpha_unfilter(1, desc, vp(a), a.size, vp(out), out.size))
        assert np.array_equal(out, oracle.unfilter_alpha(a, w, f)), (w, h, f)


def test_host_pointer_halves_recon_and_loopfilter():
    """wp_vp8_recon (unfiltered planes) and wp_vp8_loopfilter (in place on row-major planes), SURVEY 8b items 2-3."""
    from oracle import oracle
    from packed_util import random_batch
    from webp_amd._lib import check, lib
    rb = random_batch(12, 99, max_w=7, max_h=9)
    y = np.zeros(rb.y_bytes, np.uint8); u = np.zeros(rb.u_bytes, np.uint8); v = np.zeros(rb.v_bytes, np.uint8)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8_recon(rb.n, rb.desc, vp(rb.mb), rb.n_mb, vp(rb.coeff), rb.n_coeff_blocks, vp(y), vp(u), vp(v)))
    exp = [rb.expected(i) for i in range(rb.n)]
    for i, e in enumerate(exp):
        d = rb.desc[i]
        nmb = d.mbw * d.mbh
        assert np.array_equal(y[d.y_off:d.y_off + 256 * nmb], e["y0"]) and np.array_equal(u[d.u_off:d.u_off + 64 * nmb], e["u0"])
    check(lib().wp_vp8_loopfilter(rb.n, rb.desc, vp(rb.filt), rb.n_mb, vp(y), vp(u), vp(v)))
    for i, e in enumerate(exp):
        d = rb.desc[i]
        nmb = d.mbw * d.mbh
        assert np.array_equal(y[d.y_off:d.y_off + 256 * nmb], e["y"]), i
        assert np.array_equal(u[d.u_off:d.u_off + 64 * nmb], e["u"]) and np.array_equal(v[d.v_off:d.v_off + 64 * nmb], e["v"]), i
