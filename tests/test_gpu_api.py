"""End-to-end GPU parity through the host-side mirror of the reference API
(WebPDecoder.decode / Tools.toArgb, per-frame conversion as test/Visual.hx): every golden image the
reference's tests hold, plus the fixtures without a golden against the CPU oracle."""
import glob
import os
import numpy as np
import pytest
from conftest import golden_input, golden_expected, GOLDEN_NAMES, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_decode_matches_reference_golden(name):
    from webp_amd import WebPDecoder, Tools
    img = WebPDecoder.decode(golden_input(name))
    frames = Tools.framesToArgb(img)
    exp = golden_expected(name)
    assert len(frames) == len(exp)
    for k, (f, e) in enumerate(zip(frames, exp)):
        assert f.size == e.size, (name, k)
        assert np.array_equal(f, e.reshape(-1)), "%s frame %d differs from test/expected" % (name, k)


def test_toargb_passthrough_and_types():
    from webp_amd import WebPDecoder, Tools, Argb, Anim, Yuv420
    ll = WebPDecoder.decode(golden_input("1ll"))
    assert isinstance(ll.data, Argb) and Tools.toArgb(ll) is ll                 # Tools.hx:8-9
    an = WebPDecoder.decode(golden_input("1a"))
    assert isinstance(an.data, Anim) and Tools.toArgb(an) is an                 # Tools.hx:45-46
    assert an.header.width == 46 and an.header.height == 32 and len(an.data.frames) == 108
    lo = WebPDecoder.decode(golden_input("1"))
    assert isinstance(lo.data, Yuv420) and lo.data.ystride == 16 * 35 and lo.data.uvstride == 8 * 35
    out = Tools.toArgb(lo)
    assert isinstance(out.data, Argb) and out.data.stride == 4 * 550 and out.header.width == 550
    # the lossy still carries the VP8 frame header (WebPDecoder.hx:236-241; Vp8Decoder.hx:126-184), copied by toArgb (Tools.hx:41-44)
    h = lo.header
    assert (h.keyFrame, h.showFrame, h.width, h.height, h.xScale, h.yScale) == (True, True, 550, 368, 0, 0)
    assert h.firstPartitionLen > 0 and h.versionNumber in (0, 1, 2, 3) and vars(out.header) == vars(h)


def test_decode_many_bad_file_fails_alone():
    """One device batch for a list of files; the file that does not parse holds the reference's error, the rest decode."""
    from webp_amd import decode_many, Tools, WebPError
    datas = [golden_input("1"), b"RIFX\x00\x00\x00\x00WEBP", golden_input("2ll"), golden_input("2")]
    got = decode_many(datas, threads=2)
    assert isinstance(got[1], WebPError) and "Missing RIFF chunk header" in str(got[1])
    for i, name in ((0, "1"), (2, "2ll"), (3, "2")):
        assert np.array_equal(Tools.framesToArgb(got[i])[0], golden_expected(name)[0].reshape(-1)), name


@pytest.mark.parametrize("name", ["a", "1x"])
def test_fixtures_without_golden_match_oracle(name):
    from oracle import oracle
    from webp_amd import WebPDecoder, Tools
    data = golden_input(name)
    want = oracle.decode(data)
    got = Tools.framesToArgb(WebPDecoder.decode(data))
    wf = want["frames"] if want["kind"] == "anim" else [want]
    assert len(got) == len(wf)
    for g, w in zip(got, wf):
        assert np.array_equal(g, w["to_argb"])


def test_bench_data_matches_oracle():
    """Synthetic streams (normal filter at real sizes, sharpness > 0, simple filter, alpha, odd sizes,
    4K, lossless with predictor + cross-colour + subtract-green, palette): parity is HIP vs oracle
    (no reference golden exists for these: SURVEY 8c)."""
    from oracle import oracle
    from webp_amd import parse, decode_batch, Tools
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", "*.webp")))
    pick = paths                                                              # every stream the bench cycles, and the rest
    datas = [open(p, "rb").read() for p in pick]
    imgs = decode_batch([parse(d) for d in datas])
    for p, d, img in zip(pick, datas, imgs):
        want = oracle.decode(d)
        got = Tools.framesToArgb(img)
        assert np.array_equal(got[0], want["to_argb"]), os.path.basename(p)
        if want["kind"] == "yuv420":
            assert np.array_equal(img.data.y, want["y"]) and np.array_equal(img.data.u, want["u"]) and np.array_equal(img.data.v, want["v"])
            if want["a"] is not None:
                assert np.array_equal(img.data.a, want["a"])


def test_errors_surface_reference_messages():
    from webp_amd import WebPDecoder, WebPError
    with pytest.raises(WebPError, match="Missing RIFF chunk header"):
        WebPDecoder.decode(b"RIFX\x00\x00\x00\x00WEBP")
    with pytest.raises(WebPError, match="no VP8/VP8L chunk found"):
        WebPDecoder.decode(b"RIFF\x04\x00\x00\x00WEBP")


@pytest.mark.parametrize("name", ["1", "4", "a"])
def test_corrupted_token_data_decodes_like_the_oracle(name):
    """Byte flips inside the token partitions still parse (the boolean decoder never fails before EOF) but yield arbitrary
    modes and coefficients: the GPU path must agree with the oracle on them bit for bit -- or both must reject."""
    from oracle import oracle
    from webp_amd import WebPDecoder, Tools, WebPError
    rng = np.random.default_rng(100 + sum(map(ord, name)))
    d = np.frombuffer(golden_input(name), np.uint8)
    checked = 0
    for it in range(10):
        m = d.copy()
        pos = rng.integers(int(0.4 * m.size), m.size, 3)
        m[pos] = rng.integers(0, 256, 3)
        data = m.tobytes()
        try:
            want = oracle.decode(data)
        except Exception:
            want = None
        try:
            got = Tools.framesToArgb(WebPDecoder.decode(data))
        except WebPError:
            got = None
        assert (want is None) == (got is None), "iteration %d: one side rejected the stream" % it
        if want is not None:
            exp = want["to_argb"] if "to_argb" in want else None
            if exp is not None:
                assert np.array_equal(got[0], exp), "iteration %d" % it
                checked += 1
    assert checked > 0


def test_streaming_decode_equals_per_file_decode():
    """webp_amd.streaming.LossyStream (chunked: threaded parse | pinned gather + H2D | kernels | D2H) returns for every
    file what the one-file path returns, across chunk boundaries and mixed sizes."""
    import glob
    import os
    import torch
    from conftest import ROOT
    from webp_amd import WebPDecoder, Tools
    from webp_amd.streaming import LossyStream
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy_*.webp"))) + \
        sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy1080_q*_[0-1].webp"))) + \
        [os.path.join(ROOT, "tests", "golden", "inputs", n + ".webp") for n in ("1", "2", "3")]
    datas = [open(p, "rb").read() for p in paths] * 2
    argb, items, y, u, v = LossyStream(threads=4, chunk=5).decode(datas)
    torch.cuda.synchronize()
    assert len(items) == len(datas)
    from oracle import oracle
    for i in (0, 1, 2, 4, 5, 6, len(datas) // 2, len(datas) - 1):
        want = oracle.decode(datas[i])                      # the CPU restatement of the reference, not another HIP path
        it = items[i]
        for name, slab in (("y", y), ("u", u), ("v", v)):
            w = want[name]
            assert np.array_equal(slab[it[name + "_off"]:it[name + "_off"] + w.size].numpy(), w), (paths[i % len(paths)], name)
        w = want["to_argb"]
        assert np.array_equal(argb[it["argb_off"]:it["argb_off"] + w.size].numpy(), w), paths[i % len(paths)]


def _raw_alpha_still(src: bytes, filt: int, seed: int) -> bytes:
    """A VP8X still made of `src`'s VP8 chunk and an uncompressed ALPH chunk (compression 0) with the given filter."""
    import struct
    pos, vp8, vp8x = 12, None, None
    while pos + 8 <= len(src):
        tag, n = src[pos:pos + 4], struct.unpack("<I", src[pos + 4:pos + 8])[0]
        if tag == b"VP8 ":
            vp8 = src[pos + 8:pos + 8 + n]
        if tag == b"VP8X":
            vp8x = src[pos + 8:pos + 8 + n]
        pos += 8 + n + (n & 1)
    w = 1 + int.from_bytes(vp8x[4:7], "little")
    h = 1 + int.from_bytes(vp8x[7:10], "little")
    rng = np.random.default_rng(seed)
    plane = (rng.integers(0, 256, w * h, dtype=np.uint8) // (1 if filt == 0 else 16)).astype(np.uint8).tobytes()
    chunk = lambda t, d: t + struct.pack("<I", len(d)) + d + (b"\0" if len(d) & 1 else b"")
    body = b"WEBP" + chunk(b"VP8X", vp8x) + chunk(b"ALPH", bytes([filt << 2]) + plane) + chunk(b"VP8 ", vp8)
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_streaming_decode_of_stills_with_alpha():
    """Stills with an ALPH chunk through LossyStream: VP8L-coded alpha (golden a.webp, libwebp's 720p streams with the
    horizontal filter) and uncompressed alpha with every filter, mixed with opaque stills in one call; A,R,G,B and the
    alpha plane itself (ImageData.Yuv420.a, WebPDecoder.hx:236-241) against the oracle."""
    import glob
    import os
    import torch
    from conftest import ROOT
    from oracle import oracle
    from webp_amd.streaming import LossyStream
    rd = lambda p: open(p, "rb").read()
    coded = [rd(os.path.join(ROOT, "tests", "golden", "inputs", "a.webp"))] + [rd(p) for p in sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy720_alpha_*.webp")))]
    raw = [_raw_alpha_still(coded[1], f, 10 + f) for f in range(4)]
    opaque = [rd(os.path.join(ROOT, "tests", "golden", "inputs", n + ".webp")) for n in ("1", "2")]
    datas = [opaque[0], coded[0], raw[1], coded[1], opaque[1], raw[3], raw[0], coded[2], raw[2]]
    argb, items, y, u, v = LossyStream(threads=4, chunk=4).decode(datas)
    torch.cuda.synchronize()
    for i, data in enumerate(datas):
        want, it = oracle.decode(data), items[i]
        w = want["to_argb"]
        assert np.array_equal(argb[it["argb_off"]:it["argb_off"] + w.size].numpy(), w), i
        if want["a"] is None:
            assert it["alpha"] is None
        else:
            assert np.array_equal(it["alpha"], want["a"]), i
        assert np.array_equal(y[it["y_off"]:it["y_off"] + want["y"].size].numpy(), want["y"]), i
