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
