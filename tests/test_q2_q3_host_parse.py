"""Host-parse parity for the two reference quirks that only show in the frame header:
  Q2  negative segment quantiser / filter-strength values and loop-filter deltas, written through the unsigned Int8Vector
      (Vp8Decoder.hx:213-218, 236-241) and read back as 256 + v (:248-252, 398-400, 501-512);
  Q3  more than one DCT partition: every partition is sliced from offset 0 of the same buffer (:260-287).
CPU: the C++ host parser's packed buffers through the oracle's packed-buffer hook against the oracle's own bitstream decode
(the restatement of the Haxe text), on hand-authored frame headers (tests/vp8_forge.py: a boolean encoder writes the header
fields, the rest of the stream is seeded random bits) and on libwebp streams encoded with 2, 4 and 8 partitions
(tools/enc_webp.c `partitions` = 1, 2, 3; bench_data/lossy_parts*.webp).
Parity unpinned by a golden: no fixture of the reference uses negative values or several partitions, so this is
restatement-vs-host-parser, backed by the line-by-line reading of the Haxe text cited above."""
import glob
import os

import numpy as np
import pytest
from conftest import ROOT
from vp8_forge import CASES, forge


def _both(data):
    from oracle import oracle
    from webp_amd import WebPError, parse
    try:
        ora = oracle.decode(data, keep_unfiltered=True)
    except oracle.OracleError as e:
        ora = str(e)
    try:
        f = parse(data).frames[0]
    except WebPError as e:
        f = e.msg
    return ora, f


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_forged_headers(name, kw):
    from oracle import oracle
    ora, f = _both(forge(**kw))
    if isinstance(ora, str) or isinstance(f, str):
        assert isinstance(ora, str) and isinstance(f, str), "%s: oracle %r, host parser %r" % (name, ora if isinstance(ora, str) else "ok", f if isinstance(f, str) else "ok")
        assert f in ora                            # same message (the oracle's carries its status code in front)
        assert name == "partition_runs_dry"
        return
    assert name != "partition_runs_dry"
    got = oracle.vp8_from_packed(f.mbw, f.mbh, f.buf(0, np.uint32), f.buf(2, np.int16), f.buf(1, np.uint32), f.filter_type)
    for k in ("y0", "u0", "v0", "y", "u", "v"):
        assert np.array_equal(got[k], ora[k]), "%s plane %s" % (name, k)


def test_negative_values_really_read_back_unsigned():
    """The point of Q2, checked on the filter records the host emits: segment strengths [-10, 5, -63, 20] relative to level
    30 with sharpness 2 give levels min(63, 30 + 246), 35, min(63, 30 + 193), 50 -- the reference's unsigned read-back --
    i.e. stored records 2 * level + ilevel = 133, 77, 133, 107; a signed reading would give 20, 35, 0 (off), 50."""
    from webp_amd import parse
    kw = dict(CASES)["rel_neg_strength"]
    f = parse(forge(**kw)).frames[0]
    levels = set(int(v) for v in np.unique(f.buf(1, np.uint32) & 0xff))
    assert levels == {77, 107, 133}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy_parts*.webp"))), ids=os.path.basename)
def test_libwebp_multi_partition_streams(path):
    """Valid streams with 2 / 4 / 8 partitions.  The reference decodes them WRONGLY (Q3) -- and so must we."""
    from oracle import oracle
    data = open(path, "rb").read()
    ora, f = _both(data)
    assert not isinstance(ora, str) and not isinstance(f, str)
    got = oracle.vp8_from_packed(f.mbw, f.mbh, f.buf(0, np.uint32), f.buf(2, np.int16), f.buf(1, np.uint32), f.filter_type)
    for k in ("y0", "u0", "v0", "y", "u", "v"):
        assert np.array_equal(got[k], ora[k]), "%s plane %s" % (os.path.basename(path), k)


def test_multi_partition_streams_exist():
    assert len(glob.glob(os.path.join(ROOT, "bench_data", "lossy_parts*.webp"))) == 3
