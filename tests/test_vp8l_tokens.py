"""Host half of SURVEY 8f item 2 (first slice): the token stream the host parse leaves when the pixel loop of decodePix is
to run on the device (include/webp_amd.h, wp_vp8l_tokens_frame).  CPU only: the stream, acted on by the oracle's
restatement of that loop (Vp8LDecoder.hx:254-333), must give the pixels the host-resolved parse holds, and the two parse
modes must fail alike.  The device kernel is checked in tests/test_gpu_vp8l.py."""
import ctypes
import glob
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOSSLESS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "*ll.webp"))) + \
           sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossless1080_*.webp")))[:2]


def _read(p):
    with open(p, "rb") as f:
        return f.read()


@pytest.mark.parametrize("path", LOSSLESS, ids=os.path.basename)
def test_token_stream_resolves_to_the_host_pixels(path):
    from oracle import oracle
    from webp_amd._lib import Vp8lTokensFrame, check, lib
    from webp_amd.decoder import parse
    data = _read(path)
    ref = parse(data).frames[0]
    tk = parse(data, vp8l_tokens=True).frames[0]
    assert tk.buf(3).size == 0 and ref.buf(6).size == 0
    tokens = tk.buf(6, np.uint32)
    assert tokens.size > 0
    assert tk.transforms == ref.transforms and tk.coded_width == ref.coded_width
    for k in range(len(ref.transforms)):
        assert np.array_equal(tk.buf(10 + k), ref.buf(10 + k))
    want = ref.buf(3)
    got = oracle.resolve_tokens(ref.coded_width, ref.vp8l_height, tk.cc_bits, tokens)
    assert np.array_equal(got, want)
    d = Vp8lTokensFrame()
    d.coded_width, d.height, d.cc_bits, d.n_words = ref.coded_width, ref.vp8l_height, tk.cc_bits, tokens.size
    check(lib().wp_vp8l_check_tokens(ctypes.byref(d), tokens.ctypes.data_as(ctypes.c_void_p)))


def test_check_tokens_refuses_what_decodepix_refuses():
    from webp_amd._lib import Vp8lTokensFrame, WebPError, check, lib
    L = lib()

    def run(w, h, cc, words):
        t = np.array(words, np.uint32)
        d = Vp8lTokensFrame()
        d.coded_width, d.height, d.cc_bits, d.n_words = w, h, cc, t.size
        check(L.wp_vp8l_check_tokens(ctypes.byref(d), t.ctypes.data_as(ctypes.c_void_p)))

    LIT, COPY, CACHE = 0 << 30, 1 << 30, 2 << 30
    run(4, 1, 0, [LIT | 1, 0x11223344, COPY | 3, 1])
    run(4, 1, 2, [LIT | 2, 1, 2, CACHE | 3, CACHE | 0])
    for words, cc, msg in (([COPY | 2, 1, LIT | 2, 0, 0], 0, "Invalid LZ77 parameters"),          # copies before the first pixel
                           ([LIT | 1, 0, COPY | 4, 1], 0, "Invalid LZ77 parameters"),              # runs past the image
                           ([LIT | 1, 0, COPY | 3, 2], 0, "Invalid LZ77 parameters"),              # source before the image
                           ([LIT | 1, 0, COPY | 3, 0], 0, "Invalid LZ77 parameters"),
                           ([LIT | 3, 0, 0, 0, CACHE | 4], 2, "Invalid color cache index"),
                           ([LIT | 3, 0, 0, 0, CACHE | 0], 0, "null color cache"),
                           ([LIT | 5, 0, 0, 0, 0, 0], 0, "more pixels"),
                           ([LIT | 4, 0, 0], 0, "past the end"),
                           ([LIT | 3, 0, 0, 0], 0, "fewer pixels"),
                           ([LIT | 3, 0, 0, 0, 3 << 30], 0, "unknown token")):
        with pytest.raises(WebPError) as e:
            run(4, 1, cc, words)
        assert msg in e.value.msg, (words, e.value.msg)


def test_both_parse_modes_fail_alike_on_truncated_files():
    from webp_amd._lib import WebPError
    from webp_amd.decoder import parse
    data = _read(LOSSLESS[0])
    for cut in (len(data) // 3, len(data) // 2, len(data) - 7):
        bad = bytearray(data[:cut])
        bad[4:8] = (len(bad) - 8).to_bytes(4, "little")           # RIFF and chunk sizes that match the shorter file
        bad[16:20] = (len(bad) - 20).to_bytes(4, "little")
        msgs = []
        for tokens in (False, True):
            with pytest.raises(WebPError) as e:
                parse(bytes(bad), vp8l_tokens=tokens)
            msgs.append(e.value.msg)
        assert msgs[0] == msgs[1]
