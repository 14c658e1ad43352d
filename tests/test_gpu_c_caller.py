"""The C ABI driven from plain C: tests/c_caller/c_caller.c includes include/webp_amd.h as C99 (-pedantic), parses 1.webp
with the library's host side, decodes it with the host-pointer entry point and dumps the planes and ARGB, which must equal
the reference's golden image / the oracle's planes.  The binary is built by __graft_entry__.build() on the CPU."""
import os
import subprocess
import numpy as np
import pytest
from conftest import ROOT, golden_input, golden_expected

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["1", "4"])
def test_plain_c_program_decodes_like_the_reference(name, tmp_path):
    from oracle import oracle
    exe = os.path.join(ROOT, "tests", "c_caller", "c_caller")
    assert os.path.exists(exe), "tests/c_caller/c_caller is not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = tmp_path / "out.bin"
    src = os.path.join(ROOT, "tests", "golden", "inputs", name + ".webp")
    r = subprocess.run([exe, src, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, np.uint8)
    w, h, mbw, mbh = np.frombuffer(raw[:16].tobytes(), np.int32)
    want = oracle.decode(golden_input(name))
    assert (w, h) == (want["width"], want["height"])
    o = 16
    for plane, size in (("y", 256 * mbw * mbh), ("u", 64 * mbw * mbh), ("v", 64 * mbw * mbh)):
        assert np.array_equal(raw[o:o + size], want[plane]), plane
        o += size
    assert np.array_equal(raw[o:o + 4 * w * h], golden_expected(name)[0].reshape(-1))


def test_plain_c_program_pipelines_batches(tmp_path):
    """tests/c_caller/c_pipeline.c: three batches through wp_vp8_pipeline_* with device memory from the HIP C API."""
    from oracle import oracle
    exe = os.path.join(ROOT, "tests", "c_caller", "c_pipeline")
    assert os.path.exists(exe), "tests/c_caller/c_pipeline is not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = tmp_path / "out.bin"
    src = os.path.join(ROOT, "tests", "golden", "inputs", "3.webp")
    r = subprocess.run([exe, src, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(out, np.uint8)
    w, h, mbw, mbh = np.frombuffer(raw[:16].tobytes(), np.int32)
    want = oracle.decode(golden_input("3"))
    o = 16
    for b in range(3):
        for plane, size in (("y", 256 * mbw * mbh), ("u", 64 * mbw * mbh), ("v", 64 * mbw * mbh), ("to_argb", 4 * w * h)):
            assert np.array_equal(raw[o:o + size], want[plane]), "batch %d %s" % (b, plane)
            o += size
    assert o == raw.size
