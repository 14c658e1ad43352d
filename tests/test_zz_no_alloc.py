"""wp_vp8_decode_to_argb_static_dev (the entry point that allocates nothing; workspace = a static device array): its
output equals wp_vp8_decode_to_argb_dev's, bit for bit, on two calls in a row -- the second with a smaller input, which
finds the first call's tiles in the shared array -- and a batch beyond WP_STATIC_WORK_BYTES is refused.  Collected last
(file name): the round's GPU run uses `pytest -x`."""
import ctypes
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _static_call(db):
    import torch
    from webp_amd._lib import check, lib
    stream = ctypes.c_void_p(torch.cuda.current_stream(db.device).cuda_stream)
    check(lib().wp_vp8_decode_to_argb_static_dev(db.b.n, db.b.desc, db.frames.data_ptr(), db.mb.data_ptr(), db.filt.data_ptr(),
                                                 db.coeff.data_ptr(), db.y.data_ptr(), db.u.data_ptr(), db.v.data_ptr(),
                                                 None, db.argb.data_ptr(), stream))


def _outputs(db):
    import torch
    torch.cuda.synchronize()
    return [t.cpu().numpy().copy() for t in (db.y, db.u, db.v, db.argb)]


def test_static_workspace_entry_point_matches_the_workspace_one():
    import torch
    from packed_util import random_batch
    from webp_amd._lib import check, lib
    from webp_amd.device import DeviceLossyBatch
    # first call: 700 frames (more than the 512 persistent workgroups), ~20 MB of tiles; second call: 40 frames
    for n, seed in ((700, 31), (40, 32)):
        rb = random_batch(n, seed, extreme=(n == 40), max_w=12, max_h=9)
        db = DeviceLossyBatch(rb, "cuda:0")
        assert db.work_bytes <= 64 << 20
        for t in (db.y, db.u, db.v, db.argb):
            t.zero_()
        db.decode_to_argb()                       # the existing entry point, caller workspace
        want = _outputs(db)
        for t in (db.y, db.u, db.v, db.argb):
            t.zero_()
        # "allocates nothing" at the level of the driver, not only of torch's caching allocator (which a hipMalloc inside the
        # library would never show up in): free device memory as the driver reports it, before and after the call, with the
        # library's one-time set-up (wp_init: side stream, the static array's address) done beforehand
        check(lib().wp_init(0))
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        mem_before = torch.cuda.memory_allocated()
        _static_call(db)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert torch.cuda.memory_allocated() == mem_before
        assert free_after == free_before, "free device memory moved by %d bytes across the static call" % (free_before - free_after)
        got = _outputs(db)
        for name, g, w in zip("yuva", got, want):
            assert np.array_equal(g, w), "n=%d: output %s of the static-workspace call differs" % (n, name)
        for i in (0, n // 2, n - 1):              # and both equal the oracle
            exp = rb.expected(i)
            d = rb.desc[i]
            assert np.array_equal(got[0][d.y_off:d.y_off + 256 * d.mbw * d.mbh], exp["y"])
            assert np.array_equal(got[3][d.argb_off:d.argb_off + 4 * d.width * d.height], exp["argb"])


def test_static_workspace_refuses_batches_beyond_its_size():
    from webp_amd._lib import Vp8Frame, WebPError, check, lib
    n = 12                                        # 12 x 1080p: 76 MB of tiles > 64 MiB
    desc = (Vp8Frame * n)()
    mb = 0
    for d in desc:
        d.width, d.height, d.mbw, d.mbh, d.filter_type, d.mb_base = 1920, 1080, 120, 68, 2, mb
        mb += 120 * 68
    work = ctypes.c_uint64(0)
    check(lib().wp_vp8_plan(n, desc, ctypes.byref(work)))
    assert work.value > 64 << 20
    with pytest.raises(WebPError, match="WP_STATIC_WORK_BYTES") as e:
        check(lib().wp_vp8_decode_to_argb_static_dev(n, desc, 1, 1, 1, 1, None, None, None, None, None, None))
    assert e.value.code == -7
