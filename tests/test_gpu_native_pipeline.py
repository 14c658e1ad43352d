"""wp_vp8_pipeline_* (the C ABI's software pipeline, csrc/pipeline.cpp): batches submitted back to back -- each one's
finishing pass running beside the next one's reconstruction + loop filter kernel, workspaces rotating -- must leave exactly
the outputs wp_vp8_decode_to_argb_dev leaves, i.e. the oracle's."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(db, rb, which):
    for i in which:
        exp = rb.expected(i)
        y, u, v = db.planes(i)
        assert np.array_equal(y, exp["y"]) and np.array_equal(u, exp["u"]) and np.array_equal(v, exp["v"]), "planes of frame %d" % i
        assert np.array_equal(db.argb_of(i), exp["argb"]), "ARGB of frame %d" % i


@pytest.mark.parametrize("workspaces", [2, 3])
def test_pipelined_batches_equal_the_oracle(workspaces):
    import torch
    from packed_util import random_batch
    from webp_amd.device import DeviceLossyBatch, NativePipeline
    sizes = (600, 40, 300, 7, 520)                       # more and fewer frames than persistent workgroups, in turn
    rbs = [random_batch(n, 400 + k, extreme=(k == 1), max_w=10, max_h=7) for k, n in enumerate(sizes)]
    dbs = [DeviceLossyBatch(rb, "cuda:0") for rb in rbs]
    pipe = NativePipeline("cuda:0", work_bytes=max(db.work_bytes for db in dbs), workspaces=workspaces, timing=True)
    for rounds in range(2):                              # the second round re-uses every workspace
        for db in dbs:
            for t in (db.y, db.u, db.v, db.argb):
                t.zero_()
        for db in dbs:
            pipe.submit(db)
        pipe.drain()
        for db, rb in zip(dbs, rbs):
            _check(db, rb, sorted(set([0, rb.n // 2, rb.n - 1])))
    ms_mb, ms_fin, n = pipe.stats()
    assert n == 2 * len(sizes) and ms_mb > 0 and ms_fin > 0
    pipe.close()


def test_flush_orders_the_callers_stream_and_small_workspaces_are_refused():
    import torch
    from packed_util import random_batch
    from webp_amd._lib import WebPError
    from webp_amd.device import DeviceLossyBatch, NativePipeline
    rb = random_batch(64, 77, max_w=8, max_h=8)
    db = DeviceLossyBatch(rb, "cuda:0")
    pipe = NativePipeline("cuda:0", work_bytes=db.work_bytes, workspaces=2)
    db.argb.zero_()
    pipe.submit(db)
    pipe.flush()                                         # no host sync: the copy below is ordered behind the step on the current stream
    out = db.argb.clone()
    torch.cuda.synchronize()
    d = rb.desc[5]
    assert np.array_equal(out[d.argb_off:d.argb_off + 4 * d.width * d.height].cpu().numpy(), rb.expected(5)["argb"])
    pipe.close()
    small = NativePipeline("cuda:0", work_bytes=4096, workspaces=2)
    with pytest.raises(WebPError, match="workspace too small"):
        small.submit(db)
    small.close()


def test_shutdown_and_reinitialise():
    """wp_shutdown destroys the library's side stream and events; the next call re-initialises."""
    from packed_util import random_batch
    from webp_amd._lib import check, lib
    from webp_amd.device import DeviceLossyBatch
    check(lib().wp_shutdown())
    rb = random_batch(5, 78, max_w=4, max_h=4)
    db = DeviceLossyBatch(rb, "cuda:0")
    db.decode_to_argb()
    import torch
    torch.cuda.synchronize()
    assert np.array_equal(db.argb_of(2), rb.expected(2)["argb"])
    check(lib().wp_shutdown())
    check(lib().wp_init(0))
