"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/webp_amd.h
declares, and the C++ host side (container walk + entropy decode) reports the same structure and the
same error strings as the reference for its own fixtures.  No compute entry point is called."""
import ctypes
import os
import re
import numpy as np
import pytest
from conftest import golden_input, ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "webp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from webp_amd import _lib
    L = ctypes.CDLL(_lib.SO_PATH)
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "missing export: " + n


def test_struct_layouts_match_header():
    from webp_amd._lib import Vp8Frame, Vp8lFrame, Vp8lTransform, AlphaFrame
    assert ctypes.sizeof(Vp8Frame) == 96 and ctypes.sizeof(Vp8lTransform) == 32
    assert ctypes.sizeof(Vp8lFrame) == 16 + 4 * 32 + 24 and ctypes.sizeof(AlphaFrame) == 32


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from webp_amd import WebPDecoder, WebPError
    with pytest.raises(WebPError, match="no HIP device"):
        WebPDecoder.decode(golden_input("1"))


@pytest.mark.parametrize("name,kind,w,h,nframes", [
    ("1", 0, 550, 368, 1), ("3", 0, 1280, 720, 1), ("1ll", 1, 400, 301, 1), ("3ll", 1, 800, 600, 1),
    ("1a", 2, 46, 32, 108), ("2a", 2, 400, 400, 12), ("a", 0, 300, 300, 1)])
def test_host_parse_structure(name, kind, w, h, nframes):
    from webp_amd import parse
    p = parse(golden_input(name))
    assert p.kind == kind and len(p.frames) == nframes
    if name != "2a":
        assert (p.width, p.height) == (w, h)
    f = p.frames[0]
    if f.type == 0:
        assert f.mbw == (f.width + 15) >> 4 or kind == 2
        mb = f.buf(0, np.uint32).reshape(-1, 4)
        assert mb.shape[0] == f.mbw * f.mbh and f.buf(1, np.uint32).size == f.mbw * f.mbh
        nz = np.array([bin(int(v) & 0x1FFFFFF).count("1") for v in mb[:, 2]])
        assert f.buf(2, np.int16).size == 16 * nz.sum()                 # one stored block per mask bit
        assert np.array_equal(mb[:, 3], np.concatenate([[0], np.cumsum(nz)[:-1]]))


def test_host_parse_matches_oracle_dimensions_for_all_frames():
    from oracle import oracle
    from webp_amd import parse
    for name in ("1a", "2a"):
        p = parse(golden_input(name))
        o = oracle.decode(golden_input(name))
        assert [(f.width, f.height, f.x, f.y, f.duration) for f in p.frames] == \
               [(f["width"], f["height"], f["x"], f["y_off"], f["duration"]) for f in o["frames"]]


def test_fixture_filter_records():
    """1.webp: simple filter, level 4; a.webp: normal filter."""
    from webp_amd import parse
    f = parse(golden_input("1")).frames[0]
    assert f.filter_type == 1
    a = parse(golden_input("a")).frames[0]
    assert a.filter_type == 2 and a.has_alpha == 1 and a.alpha_is_vp8l == 1


@pytest.mark.parametrize("data,msg", [
    (b"RIFX\x00\x00\x00\x00WEBP", "Missing RIFF chunk header"),
    (b"RIFF\x02\x00\x00\x00WEBP", "Short chunk data"),
    (b"RIFF\x04\x00\x00\x00WEBP", "Invalid format: no VP8/VP8L chunk found"),
    (b"RIFF\x10\x00\x00\x00WEBPVP8L\x04\x00\x00\x00\x2e\x00\x00\x00", "vp8l: invalid header"),
    (b"RIFF\x16\x00\x00\x00WEBPVP8 \x0a\x00\x00\x00\x00\x00\x00\x9d\x01\x2b\x10\x00\x10\x00", "vp8: invalid format"),
    # a bad payload FOLLOWED by a truncated chunk: decodeChunks (WebPDecoder.hx:246-256) collects every chunk before any
    # payload is decoded, so the container error comes first
    (b"RIFF\x40\x00\x00\x00WEBPVP8L\x04\x00\x00\x00\x2e\x00\x00\x00JUNK\x30\x00\x00\x00abcd", "Eof"),
])
def test_host_parse_errors_match_reference_strings(data, msg):
    from webp_amd import parse, WebPError
    with pytest.raises(WebPError) as e:
        parse(data)
    assert msg in str(e.value)
    from oracle import oracle
    with pytest.raises(oracle.OracleError) as eo:
        oracle.decode(data)
    assert msg in str(eo.value)


def test_truncated_stream_is_null_frame():
    """A VP8 stream cut short makes the boolean decoder hit EOF: decodeFrame returns null."""
    from webp_amd import parse, WebPError
    full = golden_input("1")
    data = bytearray(full[:len(full) - 2000])
    data[4:8] = (len(data) - 8).to_bytes(4, "little")
    data[16:20] = (len(data) - 20).to_bytes(4, "little")
    with pytest.raises(WebPError) as e:
        parse(bytes(data))
    assert e.value.code == -5


def test_parse_batch_threads_match_serial_parse():
    """wp_parse_batch on several native threads gives byte-identical packed buffers, in order; a bad file fails the call
    with its index in the message and leaves nothing leaked that the wrapper does not own."""
    import glob
    from webp_amd import parse, parse_many
    from webp_amd._lib import WebPError
    datas = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "*.webp")))] * 3
    many = parse_many(datas, threads=4)
    assert len(many) == len(datas)
    for d, m in zip(datas, many):
        s = parse(d)
        assert (s.kind, s.width, s.height, len(s.frames)) == (m.kind, m.width, m.height, len(m.frames))
        for fs, fm in zip(s.frames, m.frames):
            for which in (0, 1, 2, 3, 4, 5):
                assert np.array_equal(fs.buf(which, np.uint8), fm.buf(which, np.uint8))
    bad = list(datas[:5])
    bad[3] = bad[3][:40]
    with pytest.raises(WebPError, match="file 3"):
        parse_many(bad, threads=3)


def test_vp8_plan_fills_tile_bases_and_workspace_size():
    """wp_vp8_plan needs no GPU: stile_base is the running sum of (mbw+1)*(mbh+1) and the workspace holds one 384-byte R
    tile per MB record slot plus the S tiles; bad descriptors are refused."""
    from webp_amd._lib import Vp8Frame, WebPError, check, lib
    dims = [(120, 68), (1, 1), (7, 3)]
    desc = (Vp8Frame * len(dims))()
    mb = 0
    for d, (w, h) in zip(desc, dims):
        d.width, d.height, d.mbw, d.mbh, d.mb_base = 16 * w, 16 * h, w, h, mb
        mb += w * h
    work = ctypes.c_uint64(0)
    check(lib().wp_vp8_plan(len(dims), desc, ctypes.byref(work)))
    st = 0
    for d, (w, h) in zip(desc, dims):
        assert d.stile_base == st
        st += (w + 1) * (h + 1)
    r_bytes = (384 * mb + 255) // 256 * 256
    assert work.value == r_bytes + 384 * st
    desc[1].mbw = 0
    with pytest.raises(WebPError, match="bad wp_vp8_frame at index 1"):
        check(lib().wp_vp8_plan(len(dims), desc, ctypes.byref(work)))


def test_mutated_files_are_rejected_or_parsed_without_crashing():
    """Random truncations / byte flips of the fixtures: wp_parse either returns a negative status with a message or a
    consistent object (tools/fuzz/run.sh runs the same mutations under ASan + UBSan)."""
    import glob
    from webp_amd._lib import lib
    rng = np.random.default_rng(7)
    L = lib()
    n_ok = n_bad = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "*.webp"))):
        d = np.frombuffer(open(path, "rb").read(), np.uint8)
        for it in range(40):
            m = d.copy()
            kind = it % 3
            if kind == 0:
                m = m[:rng.integers(0, m.size + 1)]
            elif kind == 1:
                m[rng.integers(0, m.size, 4)] = rng.integers(0, 256, 4)
            else:
                m[rng.integers(0, min(m.size, 64), 2)] = rng.integers(0, 256, 2)
            data = m.tobytes()
            h = ctypes.c_void_p()
            rc = L.wp_parse(data, len(data), ctypes.byref(h))
            if rc == 0:
                assert h.value and L.wp_parsed_nframes(h) >= 1 and L.wp_parsed_width(h) > 0
                L.wp_parsed_free(h)
                n_ok += 1
            else:
                assert rc < 0 and not h.value
                n_bad += 1
    assert n_ok > 0 and n_bad > 0


def test_batch_layout_and_gather():
    """wp_vp8_batch_layout / wp_vp8_batch_gather: descriptors, bases and slabs equal the per-frame buffers laid end to end;
    capacities and non-lossy items are checked."""
    import ctypes as C
    from webp_amd import parse
    from webp_amd._lib import Vp8BatchSizes, Vp8Frame, WebPError, check, lib
    from webp_amd.decoder import LossyBatch
    files = [parse(golden_input(n)) for n in ("1", "3", "a", "2")]
    frames = [f.frames[0] for f in files]
    for threads in (1, 3):
        b = LossyBatch(frames, threads=threads)
        mb = np.concatenate([f.buf(0, np.uint32) for f in frames])
        fl = np.concatenate([f.buf(1, np.uint32) for f in frames])
        co = np.concatenate([f.buf(2, np.int16) for f in frames])
        assert np.array_equal(b.mb, mb) and np.array_equal(b.filt, fl) and np.array_equal(b.coeff, co)
        nmb = st = blocks = 0
        for i, f in enumerate(frames):
            d = b.desc[i]
            assert (d.width, d.height, d.mbw, d.mbh, d.filter_type) == (f.width, f.height, f.mbw, f.mbh, f.filter_type)
            assert (d.mb_base, d.stile_base, d.coeff_base) == (nmb, st, blocks)
            assert d.y_off % 256 == 0 and d.u_off % 256 == 0 and d.argb_off % 256 == 0
            nmb += f.mbw * f.mbh
            st += (f.mbw + 1) * (f.mbh + 1)
            blocks += f.buf(2, np.int16).size // 16
        assert b.n_mb == nmb and b.n_coeff_blocks == blocks
        assert b.work_bytes == (nmb * 384 + 255) // 256 * 256 + st * 384
    L = lib()
    n = len(frames)
    hs = (C.c_void_p * n)(*[f._owner._h for f in frames])
    desc = (Vp8Frame * n)()
    sizes = Vp8BatchSizes()
    check(L.wp_vp8_batch_layout(n, hs, None, None, desc, C.byref(sizes)))
    small = np.zeros(16, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    with pytest.raises(WebPError, match="slab too small"):
        check(L.wp_vp8_batch_gather(n, hs, None, desc, p(small), p(small), 4, p(small), 1, 1))
    lossless = parse(golden_input("1ll"))
    hs2 = (C.c_void_p * 1)(lossless._h)
    with pytest.raises(WebPError, match="not a lossy frame"):
        check(L.wp_vp8_batch_layout(1, hs2, None, None, desc, C.byref(sizes)))


def test_parse_job_delivers_files_in_order_while_parsing():
    """wp_parse_job_*: a background parse whose results are taken in index order; errors surface at the chunk that holds
    the bad file; closing early leaves nothing behind."""
    import glob
    from webp_amd import WebPError, parse
    from webp_amd.decoder import ParseJob
    datas = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "*.webp")))]
    ref = [(f.kind, f.width, f.height, len(f.frames)) for f in (parse(d) for d in datas)]
    with ParseJob(datas * 2, 3) as job:
        got = job.take(4) + job.take(9) + job.take(9) + job.take(10 ** 6)
    assert [(f.kind, f.width, f.height, len(f.frames)) for f in got] == ref * 2
    bad = [datas[0], b"RIFF\x04\x00\x00\x00WEBP", datas[1], datas[2]]
    with ParseJob(bad, 2) as job:
        assert len(job.take(1)) == 1
        with pytest.raises(WebPError, match="file 1"):
            job.take(3)
    with ParseJob(datas * 4, 2) as job:          # closed with most files never taken
        job.take(2)
    with ParseJob([], 2) as job:
        assert job.take(5) == []


def test_header_is_plain_c99():
    """include/webp_amd.h compiles as C99 with -pedantic (what a HashLink .hdll shim or any other C binding includes)."""
    import subprocess, tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".c") as f:
        f.write('#include "webp_amd.h"\nint main(void) { return (int)sizeof(wp_vp8_frame) - 96; }\n')
        f.flush()
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), f.name])


def test_check_frames_reports_bad_items_alone():
    """wp_vp8_check_frames: a bad descriptor is reported by index, the others stay usable (a bad image fails alone)."""
    from webp_amd._lib import Vp8Frame, lib
    desc = (Vp8Frame * 4)()
    for i, d in enumerate(desc):
        d.width, d.height, d.mbw, d.mbh, d.filter_type = 30, 20, 2, 2, i % 3
    desc[2].width = 33                      # wider than its macroblock grid
    status = (ctypes.c_int * 4)()
    assert lib().wp_vp8_check_frames(4, desc, status) == 1
    assert list(status) == [0, 0, -1, 0]


def test_parse_many_lets_a_bad_file_fail_alone():
    from webp_amd import parse_many, ParsedFile, WebPError
    datas = [golden_input("1"), b"RIFF\x04\x00\x00\x00WEBP", golden_input("1ll"), golden_input("1")[:300]]
    got = parse_many(datas, threads=2, strict=False)
    assert isinstance(got[0], ParsedFile) and isinstance(got[2], ParsedFile)
    assert isinstance(got[1], WebPError) and "no VP8/VP8L chunk found" in str(got[1])
    assert isinstance(got[3], WebPError)


def test_plan_schedules_heavy_frames_with_light_ones():
    """wp_vp8_plan's `turn` (the one-kernel path's frame schedule): a permutation per launch, heaviest frames first, rounds
    of one frame per persistent workgroup in alternating direction; without a GPU the grid defaults to one round."""
    from webp_amd._lib import Vp8Frame, check, lib
    L = lib()
    n = 37
    desc = (Vp8Frame * n)()
    rng = np.random.default_rng(5)
    for i in range(n):
        d = desc[i]
        d.mbw, d.mbh = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        d.width, d.height = 16 * d.mbw, 16 * d.mbh
        d.cost = int(rng.integers(0, 50))               # 0 = unknown: the MB count stands in
    work = ctypes.c_uint64()
    check(L.wp_vp8_plan(n, desc, ctypes.byref(work)))
    turns = [int(desc[i].turn) for i in range(n)]
    assert sorted(turns) == list(range(n))
    cost = lambda i: int(desc[i].cost) or desc[i].mbw * desc[i].mbh
    costs = [cost(t) for t in turns]
    # within every round of `grid` slots the costs are monotonic; with no GPU (grid unknown -> 2 per CU of 0 CUs -> 1) or a
    # grid larger than the batch this is one descending run
    assert costs[0] == max(costs)


def test_anmf_header_larger_than_the_vp8_stream_does_not_fail_the_batch():
    """An ANMF frame whose header claims a larger size than its VP8 stream decodes with the stream's size (the reference sizes
    the planes from the VP8 header, Vp8Decoder.ensureImg) instead of making wp_vp8_batch_layout refuse the whole batch."""
    import glob
    import struct
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_anim import build_anim
    from webp_amd.decoder import LossyBatch, parse
    stills = sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy720_alpha_*.webp")))
    if not stills:
        pytest.skip("bench_data/lossy720_alpha_*.webp missing")
    data = bytearray(build_anim([open(p, "rb").read() for p in stills], 2))
    pos = 12
    while pos + 8 <= len(data):
        tag, n = bytes(data[pos:pos + 4]), struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == b"ANMF":
            data[pos + 8 + 6:pos + 8 + 9] = (4000 - 1).to_bytes(3, "little")        # frame width minus one: far beyond the stream's 1280
            break
        pos += 8 + n + (n & 1)
    pf = parse(bytes(data))
    assert pf.frames[0].width == 4000
    b = LossyBatch(pf.frames)
    assert (b.desc[0].width, b.desc[0].height) == (1280, 720) and (b.desc[1].width, b.desc[1].height) == (1280, 720)


def test_build_pins_the_floating_point_flags_the_kernels_rely_on():
    """The loop filter computes its edge masks on fp32 denormals and the colour conversion must not be contracted: both are
    build flags (webp_amd/csrc/Makefile).  That the built kernels behave accordingly is what the GPU parity tests show (a
    build that flushes denormals filters no edge at all)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = open(os.path.join(root, "webp_amd", "csrc", "Makefile")).read()
    flags = [l for l in mk.split("\n") if l.startswith("HIPFLAGS") or l.startswith("CXXFLAGS")]
    assert any("-fno-gpu-flush-denormals-to-zero" in l for l in flags) and any("-ffp-contract=off" in l for l in flags)
