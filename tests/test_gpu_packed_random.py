"""GPU parity on inputs no encoder produces: random packed macroblock buffers (every prediction mode at every position,
full-range int16 coefficients incl. the Q5/Q6 wrap cases, arbitrary loop-filter records incl. the Q4 clamp extremes),
at sizes that exercise partial bands, single rows / columns and the band hand-over -- against the oracle's packed-buffer
hook (validated against its own bitstream decode in tests/test_packed_hook.py).  Calls go through the C ABI."""
import ctypes
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(1, 1), (2, 1), (1, 2), (5, 9), (13, 4), (37, 2), (2, 33), (8, 8), (17, 5), (120, 3)]


def make_frame(rng, mbw, mbh, extreme):
    n = mbw * mbh
    mb = np.zeros((n, 4), np.uint32)
    blocks = []
    nblk = 0
    for i in range(n):
        i16 = rng.random() < 0.5
        w2 = 0
        present = rng.random(24) < rng.choice([0.0, 0.3, 0.6, 1.0])
        for b in range(24):
            if present[b]:
                w2 |= 1 << b
        has_wht = i16 and rng.random() < 0.8
        if has_wht:
            w2 |= 1 << 24
        if i16:
            w2 |= 1 << 25
            mb[i, 0] = rng.integers(0, 4)
        else:
            modes = rng.integers(0, 10, 16)
            mb[i, 0] = sum(int(m) << (4 * k) for k, m in enumerate(modes[:8]))
            mb[i, 1] = sum(int(m) << (4 * k) for k, m in enumerate(modes[8:]))
        w2 |= int(rng.integers(0, 4)) << 26
        mb[i, 2] = w2
        mb[i, 3] = nblk
        cnt = int(has_wht) + int(present.sum())
        for k in range(cnt):
            kind = rng.random()
            if extreme and kind < 0.15:
                c = rng.integers(-32768, 32768, 16)                      # anything an int16 can hold
            elif extreme and kind < 0.3:
                c = rng.choice([-32768, -25081, -25080, 25080, 25081, 32767, 0], 16)   # around the 85627*x wrap point
            else:
                c = np.rint(rng.laplace(0, 40, 16)).astype(np.int64)
                c[rng.random(16) < 0.6] = 0
            blocks.append(np.clip(c, -32768, 32767).astype(np.int16))
        if i16:                                                          # i16 luma blocks carry DC = 0 (the WHT supplies it)
            off = nblk + int(has_wht)
            for b in range(16):
                if present[b]:
                    blocks[off][0] = 0
                    off += 1
        nblk += cnt
    coeff = np.concatenate(blocks) if blocks else np.zeros(16, np.int16)
    level = rng.integers(0, 131, n)            # stored level = 2*level + ilevel can reach 126 + 9; (+4 on MB edges)
    level[rng.random(n) < 0.15] = 0
    ilevel = rng.integers(0, 10, n)
    hlevel = rng.integers(0, 3, n)
    inner = rng.integers(0, 2, n)
    filt = (level | (ilevel << 8) | (hlevel << 16) | (inner << 24)).astype(np.uint32)
    return mb.reshape(-1), coeff, filt


@pytest.mark.parametrize("filter_type", [0, 1, 2])
@pytest.mark.parametrize("extreme", [False, True])
def test_random_packed_frames(filter_type, extreme):
    from oracle import oracle
    from webp_amd._lib import Vp8Frame, check, lib
    rng = np.random.default_rng(1000 + 10 * filter_type + int(extreme))
    frames = [make_frame(rng, w, h, extreme) for (w, h) in SIZES]
    n = len(frames)
    desc = (Vp8Frame * n)()
    mb_base = cbase = yo = co = ao = 0
    for i, ((w, h), (mb, coeff, filt)) in enumerate(zip(SIZES, frames)):
        d = desc[i]
        d.width, d.height, d.mbw, d.mbh = 16 * w - (i % 5), 16 * h - (i % 7), w, h      # visible size: anything inside the MB grid
        d.filter_type = filter_type
        d.mb_base, d.coeff_base = mb_base, cbase
        d.y_off, d.u_off, d.v_off, d.argb_off = yo, co, co, ao
        mb_base += w * h
        cbase += coeff.size // 16
        yo += 256 * w * h
        co += 64 * w * h + (-(64 * w * h)) % 256
        yo += (-yo) % 256
        ao += 4 * d.width * d.height
        ao += (-ao) % 256
    mb_all = np.concatenate([f[0] for f in frames])
    co_all = np.concatenate([f[1] for f in frames])
    fl_all = np.concatenate([f[2] for f in frames])
    y = np.zeros(yo, np.uint8); u = np.zeros(co, np.uint8); v = np.zeros(co, np.uint8); argb = np.zeros(ao, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8_decode_to_argb(n, desc, p(mb_all), mb_base, p(fl_all), p(co_all), cbase, p(y), p(u), p(v), None, p(argb)))
    for i, ((w, h), (mb, coeff, filt)) in enumerate(zip(SIZES, frames)):
        exp = oracle.vp8_from_packed(w, h, mb, coeff, filt, filter_type)
        d = desc[i]
        nmb = w * h
        for name, got, off, sz in (("y", y, d.y_off, 256 * nmb), ("u", u, d.u_off, 64 * nmb), ("v", v, d.v_off, 64 * nmb)):
            g = got[off:off + sz]
            bad = np.flatnonzero(g != exp[name])
            stride = 16 * w if name == "y" else 8 * w
            assert bad.size == 0, "frame %d (%dx%d MBs) plane %s: %d bytes differ, first at row %d col %d (got %d want %d)" % (
                i, w, h, name, bad.size, bad[0] // stride, bad[0] % stride, g[bad[0]], exp[name][bad[0]])
        want = oracle.yuv420_to_argb(exp["y"], 16 * w, exp["u"], exp["v"], 8 * w, None, d.width, d.height)
        assert np.array_equal(argb[d.argb_off:d.argb_off + 4 * d.width * d.height], want), "frame %d ARGB" % i


def test_out_of_range_block_index_is_refused():
    """The host-pointer entry point validates the coefficient-block indices of every MB record before any launch."""
    from webp_amd._lib import Vp8Frame, WebPError, check, lib
    rng = np.random.default_rng(5)
    mb, coeff, filt = make_frame(rng, 3, 2, False)
    desc = (Vp8Frame * 1)()
    d = desc[0]
    d.width, d.height, d.mbw, d.mbh, d.filter_type = 48, 32, 3, 2, 2
    mb = mb.copy()
    mb[4 * 5 + 2] |= 1                      # make sure the last MB stores a block ...
    mb[4 * 5 + 3] = coeff.size // 16        # ... and point it past the end of the coefficient array
    y = np.zeros(256 * 6, np.uint8); u = np.zeros(64 * 6, np.uint8); v = np.zeros(64 * 6, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    with pytest.raises(WebPError, match="coefficient block index out of range in frame 0, macroblock 5"):
        check(lib().wp_vp8_decode_to_argb(1, desc, p(mb), 6, p(filt), p(coeff), coeff.size // 16, p(y), p(u), p(v), None, None))


def test_device_resident_records_cannot_reach_beyond_the_frames_blocks():
    """The *_dev entry points cannot inspect records that live in device memory.  With WP_FRAME_COEFF_BOUNDED (what
    wp_vp8_batch_layout writes) the kernels bound every coefficient fetch of a frame by its block count, so a block index that
    was corrupted ON THE DEVICE reads zeros -- the output equals the decode of the same records with those blocks zeroed --
    instead of whatever lies behind the slab.  Both lossy kernels (one-kernel path and stand-alone reconstruction)."""
    import torch
    from oracle import oracle
    from packed_util import RawBatch
    from webp_amd.device import DeviceLossyBatch
    rng = np.random.default_rng(91)
    shapes = [(5, 4), (7, 3)]
    frames = [make_frame(rng, w, h, False) for (w, h) in shapes]
    rb = RawBatch(shapes, frames, [2, 1])
    db = DeviceLossyBatch(rb, "cuda:0")
    # corrupt frame 1 on the device: MB 9 points at the frame's last block (its further blocks fall off the end), MB 12 far
    # beyond it, MB 15 1.5 GB beyond it
    nblk = frames[1][1].size // 16
    mb1 = frames[1][0].reshape(-1, 4).copy()
    counts = [bin(int(w) & 0x00ffffff).count("1") + ((int(w) >> 24) & 1) for w in mb1[:, 2]]
    with_blocks = [m for m, c in enumerate(counts) if c >= 2]
    assert len(with_blocks) >= 3
    victims = dict(zip((with_blocks[1], with_blocks[len(with_blocks) // 2], with_blocks[-1]), (nblk - 1, nblk + 12345, 0x03000000)))
    mbd = db.mb.view(torch.int32).view(-1, 4)
    base = rb.desc[1].mb_base
    for m, idx in victims.items():
        mbd[base + m, 3] = idx
    torch.cuda.synchronize()
    # what the bounded fetches amount to: out-of-range blocks are zeros
    co = frames[1][1].reshape(-1, 16)
    exp_co = np.concatenate([co, np.zeros((26 * len(victims), 16), np.int16)])
    exp_mb = mb1.copy()
    nxt = nblk
    for m, idx in victims.items():
        cnt = bin(int(mb1[m, 2]) & 0x00ffffff).count("1") + ((int(mb1[m, 2]) >> 24) & 1)
        exp_mb[m, 3] = nxt
        if idx == nblk - 1:                      # its first block is the frame's last one, the rest is out of range
            exp_co[nxt] = co[nblk - 1]
        nxt += cnt
    want = [oracle.vp8_from_packed(5, 4, frames[0][0], frames[0][1], frames[0][2], 2),
            oracle.vp8_from_packed(7, 3, exp_mb.reshape(-1), exp_co.reshape(-1), frames[1][2], 1)]
    for path in ("one_kernel", "recon_only"):
        for t in (db.y, db.u, db.v):
            t.zero_()
        if path == "one_kernel":
            db.decode_to_argb()
            key = ("y", "u", "v")
        else:
            db.recon(planes=True)
            key = ("y0", "u0", "v0")                # the unfiltered planes
        torch.cuda.synchronize()
        for i in range(2):
            got = db.planes(i)
            for name, g in zip(key, got):
                assert np.array_equal(g, want[i][name]), "%s: plane %s of frame %d" % (path, name, i)


@pytest.mark.parametrize("shape", [(1024, 5), (3, 1024), (1024, 1)])
def test_maximum_dimensions(shape):
    """The largest macroblock grid a VP8 frame header can describe in one direction (14-bit dimensions: 16383 px = 1024
    MBs): band progress words, tile indices and the finishing pass's row segments at their limits."""
    from oracle import oracle
    from webp_amd._lib import Vp8Frame, check, lib
    w, h = shape
    rng = np.random.default_rng(77 + w + 3 * h)
    mb, coeff, filt = make_frame(rng, w, h, False)
    desc = (Vp8Frame * 1)()
    d = desc[0]
    d.width, d.height, d.mbw, d.mbh, d.filter_type = min(16 * w, 16383), min(16 * h, 16383), w, h, 2
    nmb = w * h
    y = np.zeros(256 * nmb, np.uint8); u = np.zeros(64 * nmb, np.uint8); v = np.zeros(64 * nmb, np.uint8)
    argb = np.zeros(4 * d.width * d.height, np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    check(lib().wp_vp8_decode_to_argb(1, desc, p(mb), nmb, p(filt), p(coeff), coeff.size // 16, p(y), p(u), p(v), None, p(argb)))
    exp = oracle.vp8_from_packed(w, h, mb, coeff, filt, 2)
    for name, got in (("y", y), ("u", u), ("v", v)):
        assert np.array_equal(got, exp[name]), "plane %s of a %d x %d MB frame" % (name, w, h)
    want = oracle.yuv420_to_argb(exp["y"], 16 * w, exp["u"], exp["v"], 8 * w, None, d.width, d.height)
    assert np.array_equal(argb, want)
