"""CPU: the oracle's packed-buffer hook (wo_vp8_from_packed) fed with the buffers the C++ host parser produces must give
exactly the planes the oracle's own bitstream decode gives -- for every lossy fixture.  This pins the packed format
(include/webp_amd.h) and the host parser without a GPU, and validates the hook that the random-input GPU tests use."""
import glob
import os
import numpy as np
import pytest
from conftest import golden_input, ROOT

FILES = ["1", "2", "3", "4", "5", "a"]


def _check(data, name):
    from oracle import oracle
    from webp_amd import parse
    ora = oracle.decode(data, keep_unfiltered=True)
    f = parse(data).frames[0]
    got = oracle.vp8_from_packed(f.mbw, f.mbh, f.buf(0, np.uint32), f.buf(2, np.int16), f.buf(1, np.uint32), f.filter_type)
    for k in ("y0", "u0", "v0", "y", "u", "v"):
        assert np.array_equal(got[k], ora[k]), "%s plane %s" % (name, k)


@pytest.mark.parametrize("name", FILES)
def test_packed_hook_matches_bitstream_decode(name):
    _check(golden_input(name), name)


def test_packed_hook_on_bench_streams():
    for p in sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy*.webp")))[:6] + sorted(glob.glob(os.path.join(ROOT, "bench_data", "odd*.webp"))):
        _check(open(p, "rb").read(), os.path.basename(p))


def _agree(data, name):
    """Host parser and oracle must either both refuse the stream or give the same planes."""
    from oracle import oracle
    from webp_amd import parse, WebPError
    try:
        ora = oracle.decode(data, keep_unfiltered=True)
    except Exception:
        ora = None
    try:
        f = parse(data).frames[0]
    except WebPError:
        f = None
    assert (ora is None) == (f is None), "%s: oracle %s, host parser %s" % (name, "refused" if ora is None else "ok", "refused" if f is None else "ok")
    if ora is None:
        return False
    got = oracle.vp8_from_packed(f.mbw, f.mbh, f.buf(0, np.uint32), f.buf(2, np.int16), f.buf(1, np.uint32), f.filter_type)
    for k in ("y", "u", "v"):
        assert np.array_equal(got[k], ora[k]), "%s plane %s" % (name, k)
    return True


@pytest.mark.parametrize("name", ["1", "3", "5"])
def test_partition_starting_with_ff(name):
    """Partition.hx:55-85: a partition whose first byte is 0xFF starts the boolean decoder with value == range; every
    bit then reads 1 and the reference's 32-bit `bits` grows until it wraps.  The host parser's wide-window decoder has
    to follow that wrap bit for bit (found by the differential fuzzer: frame 1.webp, width byte 0xEF, first byte 0xFF)."""
    data = bytearray(golden_input(name))
    assert data[12:16] == b"VP8 "
    first_part = (data[20] | (data[21] << 8) | (data[22] << 16)) >> 5
    n_ok = 0
    for at, width_byte in ((30, None), (30, 0xEF), (30 + first_part, None), (30 + first_part, 0x10)):
        m = bytearray(data)
        m[at] = 0xFF
        if width_byte is not None:
            m[26] = width_byte
        n_ok += _agree(bytes(m), "%s@%d" % (name, at))
        for cut in (at + 1, at + 9, at + 200, len(m) // 2):
            n_ok += _agree(bytes(m[:cut]), "%s@%d cut %d" % (name, at, cut))
    assert n_ok > 0


@pytest.mark.parametrize("name", ["1", "2", "4"])
def test_mutated_streams_agree_with_oracle(name):
    """Differential check of the host parser against the oracle's bitstream decode on damaged streams (byte flips in
    the headers, in the partitions, truncation): the same streams are refused, and the accepted ones decode alike
    (decodeFrame's null on unexpectedEOF, Q11, included)."""
    d = np.frombuffer(golden_input(name), np.uint8)
    rng = np.random.default_rng(sum(map(ord, name)))
    n_ok = n_bad = 0
    for it in range(36):
        m = d.copy()
        kind = it % 4
        if kind == 0:
            m = m[:rng.integers(20, m.size)]
        elif kind == 1:
            m[rng.integers(30, m.size, 3)] = rng.integers(0, 256, 3)
        elif kind == 2:
            m[rng.integers(20, 64, 2)] = rng.integers(0, 256, 2)
        else:
            p = int(rng.integers(30, m.size - 64))
            m[p:p + 48] = rng.integers(0, 256, 48)
        if _agree(m.tobytes(), "%s mutation %d" % (name, it)):
            n_ok += 1
        else:
            n_bad += 1
    assert n_ok > 0
