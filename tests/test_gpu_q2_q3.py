"""GPU half of tests/test_q2_q3_host_parse.py: the same hand-authored headers (negative segment values / loop-filter deltas,
Q2) and multi-partition streams (Q3) through wp_parse -> the device path, against oracle.decode: planes and ARGB.
Parity unpinned by a golden (no reference fixture covers either quirk)."""
import glob
import os

import numpy as np
import pytest
from vp8_forge import CASES, forge

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forged_and_multi_partition_streams_on_the_device():
    from oracle import oracle
    from webp_amd import Tools, WebPError
    from webp_amd.decoder import decode_many
    datas = [forge(**kw) for _, kw in CASES] + [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy_parts*.webp")))]
    names = [c[0] for c in CASES] + ["parts2", "parts4", "parts8"]
    assert len(datas) == len(CASES) + 3
    images = decode_many(datas)                  # one device batch; a refused stream fails alone
    n_ok = 0
    for name, data, img in zip(names, datas, images):
        try:
            want = oracle.decode(data)
        except oracle.OracleError as e:
            assert isinstance(img, WebPError) and img.msg in str(e), name
            continue
        assert not isinstance(img, WebPError), "%s: %s" % (name, img)
        for k, got in (("y", img.data.y), ("u", img.data.u), ("v", img.data.v)):
            assert np.array_equal(got, want[k]), "%s plane %s" % (name, k)
        assert np.array_equal(Tools.toArgb(img).data.pix, want["to_argb"]), name
        n_ok += 1
    assert n_ok == len(datas) - 1                # all but the stream whose partition runs dry
