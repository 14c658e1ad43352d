"""GPU parity of wp_vp8l_resolve[_dev] (first slice): decodePix's pixel loop -- literal stores, LZ77
copies, lazily filled colour cache (Vp8LDecoder.hx:254-333) -- on the device, against the oracle's restatement of that loop
on the same token streams; and whole files decoded with the loop on the device against the oracle's decode."""
import ctypes
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIT, COPY, CACHE = 0 << 30, 1 << 30, 2 << 30


def _resolve(cases):
    """cases: [(w, h, cc_bits, tokens)] -> list of RGBA byte arrays through the host-pointer entry point (one call)."""
    from webp_amd._lib import Vp8lTokensFrame, check, lib
    n = len(cases)
    desc = (Vp8lTokensFrame * n)()
    toks, to, po = [], 0, 0
    for i, (w, h, cc, t) in enumerate(cases):
        d = desc[i]
        d.coded_width, d.height, d.cc_bits, d.tok_off, d.n_words, d.pix_off = w, h, cc, 4 * to, t.size, po
        toks.append(t)
        to += t.size
        po += (4 * w * h + 255) // 256 * 256
    tokens = np.concatenate(toks).astype(np.uint32)
    pix = np.zeros(po, np.uint8)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8l_resolve(n, desc, vp(tokens), tokens.size, vp(pix), pix.size))
    return [pix[int(desc[i].pix_off):int(desc[i].pix_off) + 4 * c[0] * c[1]] for i, c in enumerate(cases)]


def _random_stream(rng, w, h, cc_bits, style):
    """A valid stream for a w x h image.  style picks the mix: 0 short tokens of every kind, 1 long runs / long copies /
    long stretches between cache lookups (the paths that leave the 16384-pixel ring), 2 period copies (dist < len)."""
    total = w * h
    out, p = [], 0
    few = rng.integers(0, 1 << 32, 7, dtype=np.uint64).astype(np.uint32)         # a small palette, so that cache keys collide and repeat
    def literals(n):
        v = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
        pick = rng.random(n) < 0.5
        v[pick] = few[rng.integers(0, few.size, int(pick.sum()))]
        return v
    while p < total:
        r = rng.random()
        left = total - p
        if p == 0 or r < 0.35:
            n = int(min(left, rng.integers(1, 5 if style != 1 else 40000)))
            if style == 1 and rng.random() < 0.5:
                n = int(min(left, rng.integers(1, 130)))
            out.append(np.array([LIT | n], np.uint32)); out.append(literals(n)); p += n
        elif r < 0.75 or cc_bits == 0:
            if style == 2:
                dist = int(min(p, rng.choice([1, 2, 3, 5, 63, 64, 65, 127, 130, 1000])))
                n = int(min(left, rng.integers(dist, 4097)))
            elif style == 1:
                dist = int(rng.integers(1, p + 1)) if rng.random() < 0.5 else int(min(p, rng.integers(12000, 40000)))
                n = int(min(left, rng.integers(1, 4097)))
            else:
                dist = int(min(p, rng.choice([1, 2, w, w + 1, max(w - 1, 1), 2 * w, int(rng.integers(1, p + 1))])))
                n = int(min(left, rng.integers(1, 70)))
            out.append(np.array([COPY | n, dist], np.uint32)); p += n
        else:
            k = int(min(left, rng.integers(1, 4)))                       # a few lookups in a row
            out.append((CACHE | rng.integers(0, 1 << cc_bits, k)).astype(np.uint32)); p += k
    return np.concatenate(out)


def _check(cases):
    from oracle import oracle
    got = _resolve(cases)
    for i, (w, h, cc, t) in enumerate(cases):
        want = oracle.resolve_tokens(w, h, cc, t)
        if not np.array_equal(got[i], want):
            bad = int(np.flatnonzero(got[i] != want)[0]) // 4
            raise AssertionError("stream %d (%dx%d, cache bits %d): first differing pixel %d of %d" % (i, w, h, cc, bad, w * h))


def test_random_streams_all_token_kinds():
    rng = np.random.default_rng(101)
    cases = []
    for i in range(700):                      # more images than one pass of workgroups over the CUs holds
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 40))
        cc = int(rng.integers(0, 12)) if i % 4 else 0
        cases.append((w, h, cc, _random_stream(rng, w, h, cc, 0)))
    _check(cases)


def test_long_tokens_far_distances_and_long_cache_gaps():
    rng = np.random.default_rng(202)
    cases = []
    for i in range(24):
        w, h = int(rng.integers(300, 3000)), int(rng.integers(30, 120))
        cc = (0, 1, 6, 11)[i % 4]
        cases.append((w, h, cc, _random_stream(rng, w, h, cc, 1)))
    _check(cases)


def test_period_copies():
    rng = np.random.default_rng(303)
    cases = []
    for i in range(40):
        w, h = int(rng.integers(50, 700)), int(rng.integers(20, 90))
        cc = (0, 3, 10)[i % 3]
        cases.append((w, h, cc, _random_stream(rng, w, h, cc, 2)))
    # one copy longer than the bitstream can code (taken in pieces of 4096), period 3, across the whole image
    w, h = 500, 400
    cases.append((w, h, 4, np.array([LIT | 3, 0x01020304, 0xa1b2c3d4, 0x55667788, COPY | (w * h - 5), 3, CACHE | 1, CACHE | 9], np.uint32)))
    _check(cases)


def test_malformed_streams_are_refused_before_upload():
    from webp_amd._lib import WebPError
    with pytest.raises(WebPError) as e:
        _resolve([(4, 2, 0, np.array([LIT | 2, 1, 2, COPY | 6, 3], np.uint32))])
    assert "Invalid LZ77 parameters" in e.value.msg


def test_dev_entry_point_stops_at_a_bad_token():
    """The _dev entry point cannot inspect device-resident streams; the kernel must still stay inside the frame's extents:
    a stream with a copy that reaches before the image stops there, and the neighbouring frame is untouched."""
    import torch
    from oracle import oracle
    from webp_amd._lib import Vp8lTokensFrame, check, lib
    good = np.array([LIT | 3, 7, 8, 9, COPY | 5, 2], np.uint32)
    bad = np.array([LIT | 2, 1, 2, COPY | 6, 3, LIT | 1, 5], np.uint32)
    desc = (Vp8lTokensFrame * 2)()
    for i, (t, off) in enumerate(((bad, 0), (good, bad.size))):
        d = desc[i]
        d.coded_width, d.height, d.cc_bits, d.tok_off, d.n_words, d.pix_off = 4, 2, 0, 4 * off, t.size, 256 * i
    tok = torch.from_numpy(np.concatenate([bad, good]).view(np.int32)).cuda()
    dd = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).cuda()
    pix = torch.full((512,), 0xEE, dtype=torch.uint8, device="cuda")
    check(lib().wp_vp8l_resolve_dev(2, desc, dd.data_ptr(), tok.data_ptr(), pix.data_ptr(), None))
    torch.cuda.synchronize()
    out = pix.cpu().numpy()
    assert np.array_equal(out[256:288], oracle.resolve_tokens(4, 2, 0, good))
    assert np.array_equal(out[:8].view(np.uint32), [1, 2]) and np.all(out[8:256] == 0xEE) and np.all(out[288:] == 0xEE)


FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "*ll.webp"))) + \
        sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossless1080_*.webp")))


def test_files_token_streams_against_the_oracle_loop():
    from webp_amd.decoder import parse_many
    datas = [open(p, "rb").read() for p in FILES]
    cases = []
    for pf in parse_many(datas, threads=8, vp8l_tokens=True):
        f = pf.frames[0]
        cases.append((f.coded_width, f.vp8l_height, f.cc_bits, f.buf(6, np.uint32).copy()))
    _check(cases)


def test_files_decode_with_the_pixel_loop_on_the_device():
    """WebPDecoder.decode's result for lossless files, with decodePix's pixel loop, the inverse transforms and the swizzle
    all on the device, against the oracle's decode of the same bytes."""
    from oracle import oracle
    from webp_amd.decoder import decode_batch, parse_many
    datas = [open(p, "rb").read() for p in FILES]
    images = decode_batch(parse_many(datas, threads=8, vp8l_tokens=True))
    for path, data, img in zip(FILES, datas, images):
        want = oracle.decode(data)
        assert want["kind"] == "argb" and np.array_equal(img.data.pix, want["argb"]), path


def test_mixed_batch_of_tokenized_and_host_resolved_frames():
    """decode_batch on files parsed both ways at once: LosslessBatch then holds token streams for some frames and pixels for
    the others in one slab, and wp_vp8l_resolve must hand the others' bytes back untouched (it once returned uninitialised
    device memory for them).  Both kinds must equal the oracle's decode."""
    from oracle import oracle
    from webp_amd.decoder import LosslessBatch, decode_batch, parse
    paths = [os.path.join(ROOT, "tests", "golden", "inputs", n) for n in ("1ll.webp", "5ll.webp", "4ll.webp", "2ll.webp")]
    datas = [open(p, "rb").read() for p in paths]
    files = [parse(d, vp8l_tokens=(i % 2 == 0)) for i, d in enumerate(datas)]
    lb = LosslessBatch([f.frames[0] for f in files])
    assert lb.n_tok == 2 and lb.n == 4
    # the slab level: after resolve(), host-resolved frames still hold their pixels and tokenized ones hold decodePix's output
    before = lb.pix.copy()
    lb.resolve()
    for i, f in enumerate(files):
        d = lb.desc[i]
        sl = slice(int(d.pix_off), int(d.pix_off) + 4 * d.coded_width * d.height)
        if i % 2:
            assert np.array_equal(lb.pix[sl], before[sl]), "host-resolved frame %d was overwritten by wp_vp8l_resolve" % i
        else:
            want = parse(datas[i]).frames[0].buf(3)
            assert np.array_equal(lb.pix[sl], want), "tokenized frame %d" % i
    # the API level
    for img, d in zip(decode_batch(files), datas):
        want = oracle.decode(d)
        assert np.array_equal(img.data.pix, want["argb"])


@pytest.mark.parametrize("ring", [4096, 8192, 16384])
def test_every_ring_size_gives_the_same_pixels(ring):
    """The kernel's LDS ring is chosen by batch size (more images resident per CU with a smaller ring); copies that reach
    further back than the ring, and cache fills longer than it, take the slab path.  Forced to each size in turn
    (wp_debug_set_resolve_ring), the long-token / far-distance / period-copy streams and a real file must come out alike."""
    from webp_amd._lib import check, lib
    from webp_amd import parse
    check(lib().wp_debug_set_resolve_ring(ring))
    try:
        rng = np.random.default_rng(500 + ring)
        cases = []
        for i in range(18):
            w, h = int(rng.integers(300, 3000)), int(rng.integers(30, 120))
            cases.append((w, h, (0, 1, 6, 11)[i % 4], _random_stream(rng, w, h, (0, 1, 6, 11)[i % 4], 1 + i % 2)))
        for i in range(60):
            w, h = int(rng.integers(1, 90)), int(rng.integers(1, 40))
            cc = int(rng.integers(0, 12))
            cases.append((w, h, cc, _random_stream(rng, w, h, cc, 0)))
        f = parse(open(os.path.join(ROOT, "tests", "golden", "inputs", "3ll.webp"), "rb").read(), vp8l_tokens=True).frames[0]
        cases.append((f.coded_width, f.vp8l_height, f.cc_bits, f.buf(6, np.uint32)))
        _check(cases)
    finally:
        check(lib().wp_debug_set_resolve_ring(0))
