"""GPU parity at the batch shape the bench times: more frames than a launch has persistent workgroups, so that every
workgroup walks its `for (f = blockIdx.x; f < n_frames; f += gridDim.x)` loop more than once (band sequence numbers
carried over between frames, LDS tiles / rings / progress words reused, band -> wavefront phase shifted by the frame
before).  Every frame is compared with the oracle (Vp8Decoder.hx:345-371 through the packed-buffer hook).  Also: the
split into several launches the library makes beyond its band-sequence range (forced with the test hook), and the same
for the VP8L predictor's persistent loop (grid = 1024 workgroups)."""
import ctypes
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_BIG = 1160          # > 2 x 512 persistent workgroups on MI355X (2 per CU): 136 workgroups get three frames, the rest two


def _big_batch(seed, n=N_BIG):
    from packed_util import RawBatch
    from test_gpu_packed_random import make_frame
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(n):
        if rng.random() < 0.08:       # some tall frames: up to 11 bands, more than the 8 wavefronts of a workgroup
            shapes.append((int(rng.integers(1, 5)), int(rng.integers(7, 45))))
        else:                         # 1-6 MB rows: one or two bands, the second one partial
            shapes.append((int(rng.integers(1, 10)), int(rng.integers(1, 7))))
    frames = [make_frame(rng, w, h, extreme=(i % 9 == 0)) for i, (w, h) in enumerate(shapes)]
    ft = [int(rng.integers(0, 3)) for _ in range(n)]      # no filter / simple / normal, mixed in one batch
    return RawBatch(shapes, frames, ft)


def _check_all(db, rb, what, planes=True, argb=True, unfiltered=False):
    import torch
    torch.cuda.synchronize()
    y, u, v, a = (t.cpu().numpy() for t in (db.y, db.u, db.v, db.argb))
    for i in range(rb.n):
        exp = rb.expected(i)
        d = rb.desc[i]
        nmb = d.mbw * d.mbh
        if planes:
            for name, got, off, sz in (("y", y, d.y_off, 256 * nmb), ("u", u, d.u_off, 64 * nmb), ("v", v, d.v_off, 64 * nmb)):
                want = exp[name + "0"] if unfiltered else exp[name]
                assert np.array_equal(got[off:off + sz], want), "%s: frame %d (%dx%d MBs, filter %d) plane %s" % (what, i, d.mbw, d.mbh, d.filter_type, name)
        if argb:
            assert np.array_equal(a[d.argb_off:d.argb_off + 4 * d.width * d.height], exp["argb"]), "%s: frame %d ARGB" % (what, i)


@pytest.fixture(scope="module")
def big():
    from webp_amd.device import DeviceLossyBatch
    rb = _big_batch(2024)
    return rb, DeviceLossyBatch(rb, "cuda:0")


def _clear(db):
    for t in (db.y, db.u, db.v, db.argb, db.work):
        t.zero_()


def test_recon_then_loopfilter_launches(big):
    rb, db = big
    _clear(db)
    db.recon(planes=True)
    _check_all(db, rb, "wp_vp8_recon_dev", argb=False, unfiltered=True)
    db.loopfilter(planes=False)
    db.finish()
    _check_all(db, rb, "wp_vp8_recon_dev + wp_vp8_loopfilter_dev + finish")


def test_recon_filter_one_launch(big):
    rb, db = big
    _clear(db)
    db.recon_filter()
    db.finish()
    _check_all(db, rb, "wp_vp8_recon_filter_dev")


def test_decode_to_argb_dev(big):
    rb, db = big
    _clear(db)
    db.decode_to_argb()
    _check_all(db, rb, "wp_vp8_decode_to_argb_dev")


def test_lossy_pipeline_steps(big):
    """Three pipelined steps over the two workspaces; the outputs of the last step are checked (each step overwrites them)."""
    from webp_amd.device import LossyPipeline
    rb, db = big
    for one_launch in (False, True):
        _clear(db)
        db.second_workspace().zero_()
        pipe = LossyPipeline("cuda:0", one_launch=one_launch)
        for _ in range(3):
            pipe.submit(db)
        pipe.drain()
        _check_all(db, rb, "LossyPipeline(one_launch=%s)" % one_launch)


def test_many_pipelines_share_two_streams():
    """Twelve LossyPipeline objects in a row (round 1 created two streams per object and once hung with a high-priority
    finishing stream after many had been created): all use the same two process-wide streams, and every one decodes right."""
    from packed_util import random_batch
    from webp_amd.device import DeviceLossyBatch, LossyPipeline, pipeline_streams
    rb = random_batch(40, 7, max_w=6, max_h=9)
    db = DeviceLossyBatch(rb, "cuda:0")
    s0 = pipeline_streams("cuda:0")
    for k in range(12):
        pipe = LossyPipeline("cuda:0", one_launch=bool(k & 1))
        assert (pipe.s_main, pipe.s_finish) == s0
        _clear(db)
        pipe.submit(db); pipe.submit(db)
        pipe.drain()
        if k in (0, 11):
            _check_all(db, rb, "pipeline %d" % k)


def test_split_into_several_launches(big):
    """frames_per_launch < n (capi.cpp): the hook caps a launch at 300 frames, so the batch takes four launches per kernel."""
    from webp_amd._lib import check, lib
    rb, db = big
    L = lib()
    L.wp_debug_set_frames_per_launch.argtypes = [ctypes.c_int]
    check(L.wp_debug_set_frames_per_launch(300))
    try:
        _clear(db)
        db.recon(planes=False)
        db.loopfilter(planes=False)
        db.finish()
        _check_all(db, rb, "separate launches, 300 frames each")
        _clear(db)
        db.decode_to_argb()
        _check_all(db, rb, "one-launch kernel, 300 frames each")
        # the batch above was planned for ONE launch, so its frame schedule (wp_vp8_frame.turn) does not fit the four launches
        # and each runs in slot order; planned with the split in force, every launch gets a schedule of its own
        from webp_amd.device import DeviceLossyBatch
        db2 = DeviceLossyBatch(rb, "cuda:0")
        turns = [int(rb.desc[i].turn) for i in range(rb.n)]
        for first in range(0, rb.n, 300):
            assert sorted(turns[first:first + 300]) == list(range(min(300, rb.n - first)))
        db2.decode_to_argb()
        _check_all(db2, rb, "one-launch kernel, 300 frames each, scheduled per launch")
        db2.finish()                                  # the finishing pass in the shape it takes beside another kernel
        _check_all(db2, rb, "finishing pass, several chunks per workgroup")
    finally:
        check(L.wp_debug_set_frames_per_launch(0))


def test_vp8l_predictor_more_frames_than_workgroups():
    """1100 small VP8L images with a predictor (+ cross-colour, subtract-green on some): the predictor kernel's grid is
    1024 workgroups, so 76 of them take a second frame."""
    from oracle import oracle
    from test_gpu_vp8l import _run_chain, _tiles
    rng = np.random.default_rng(4242)
    cases, want = [], []
    for i in range(1100):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 150 if i % 11 == 0 else 20))
        pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
        if i % 2:
            pix = (pix // 16).astype(np.uint8)
        bits = int(rng.integers(2, 5))
        tp = _tiles(rng, w, h, bits, list(range(16)))
        kind = i % 3
        if kind == 0:
            chain = [(0, bits, w, tp)]
            e = oracle.inverse_predictor(w, bits, tp, pix, h)
        elif kind == 1:
            tc = _tiles(rng, w, h, 3)
            chain = [(2, 0, w, np.zeros(0, np.uint8)), (0, bits, w, tp), (1, 3, w, tc)]
            e = oracle.inverse_subtract_green(oracle.inverse_predictor(w, bits, tp, oracle.inverse_cross_color(w, 3, tc, pix, h), h))
        else:
            chain = [(0, bits, w, tp), (2, 0, w, np.zeros(0, np.uint8))]
            e = oracle.inverse_predictor(w, bits, tp, oracle.inverse_subtract_green(pix), h)
        cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=chain))
        want.append(np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1))
    got = _run_chain(cases, swizzle=1)
    for i, (g, e) in enumerate(zip(got, want)):
        assert np.array_equal(g, e), "frame %d (%dx%d)" % (i, cases[i]["width"], cases[i]["height"])


def test_alpha_unfilter_many_planes():
    """1200 alpha planes, all four filters, raw and green-of-RGBA sources, in one call."""
    from oracle import oracle
    from webp_amd._lib import AlphaFrame, lib, check
    rng = np.random.default_rng(77)
    n = 1200
    desc = (AlphaFrame * n)()
    srcs, so, do, want, dims = [], 0, 0, [], []
    for i in range(n):
        w, h, f, rgba = int(rng.integers(1, 70)), int(rng.integers(1, 140 if i % 13 == 0 else 12)), i % 4, (i // 4) % 2
        a = rng.integers(0, 256, w * h, dtype=np.uint8)
        if f == 3:
            a = (a // 8).astype(np.uint8)
        want.append(oracle.unfilter_alpha(a, w, f))
        if rgba:
            s = rng.integers(0, 256, 4 * w * h, dtype=np.uint8); s[1::4] = a
        else:
            s = a
        d = desc[i]
        d.width, d.height, d.filter, d.src_is_vp8l_rgba, d.src_off, d.dst_off = w, h, f, rgba, so, do
        pad = (-s.size) % 16
        srcs.append(np.concatenate([s, np.zeros(pad, np.uint8)])); so += s.size + pad
        do += (w * h + 15) // 16 * 16
        dims.append((w, h, f))
    src = np.concatenate(srcs); out = np.zeros(do, np.uint8)
    vp = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_alpha_unfilter(n, desc, vp(src), src.size, vp(out), out.size))
    for i, (w, h, f) in enumerate(dims):
        assert np.array_equal(out[desc[i].dst_off:desc[i].dst_off + w * h], want[i]), (i, w, h, f)


def test_vp8l_chains_on_two_streams():
    """A batch large enough (>= 64 Mpix) for the library to run the chains of frames with a predictor and of frames without
    on two streams (vp8l.hip wpk_launch_vp8l): 40 predictor frames and 40 palette / subtract-green frames of 1024 x 1024,
    every frame against the oracle."""
    from oracle import oracle
    from test_gpu_vp8l import _run_chain, _tiles
    rng = np.random.default_rng(808)
    w = h = 1024
    cases, want = [], []
    for i in range(80):
        pix = rng.integers(0, 256, 4 * w * h, dtype=np.uint8)
        if i % 2 == 0:
            bits = 4
            tp = _tiles(rng, w, h, bits, list(range(14)))
            chain = [(2, 0, w, np.zeros(0, np.uint8)), (0, bits, w, tp)]
            e = oracle.inverse_subtract_green(oracle.inverse_predictor(w, bits, tp, pix, h))
        elif i % 4 == 1:
            chain = [(2, 0, w, np.zeros(0, np.uint8))]
            e = oracle.inverse_subtract_green(pix)
        else:
            pal = rng.integers(0, 256, 4 * 256, dtype=np.uint8)
            chain = [(3, 0, w, pal)]
            e = oracle.inverse_color_indexing(w, 0, pal, pix, h)
        cases.append(dict(width=w, height=h, coded_width=w, pix=pix, transforms=chain))
        want.append(np.ascontiguousarray(e.reshape(-1, 4)[:, [3, 0, 1, 2]]).reshape(-1))
    got = _run_chain(cases, swizzle=1)
    for i, (g, e) in enumerate(zip(got, want)):
        assert np.array_equal(g, e), "frame %d" % i
