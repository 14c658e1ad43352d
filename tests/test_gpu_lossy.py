"""GPU parity of the lossy path (recon -> loop filter -> convert) against the CPU oracle and the
reference's golden images.  Calls go through the C ABI (_dev entry points)."""
import numpy as np
import pytest
from conftest import golden_input, golden_expected

pytestmark = pytest.mark.gpu

LOSSY = ["1", "2", "3", "4", "5"]


@pytest.fixture(scope="module")
def dev_batch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from oracle import oracle
    from webp_amd import parse
    from webp_amd.decoder import LossyBatch
    from webp_amd.device import DeviceLossyBatch
    names = LOSSY + ["a"]
    files = [parse(golden_input(n)) for n in names]
    frames = [f.frames[0] for f in files]
    ora = [oracle.decode(golden_input(n), keep_unfiltered=True) for n in names]
    db = DeviceLossyBatch(LossyBatch(frames))
    return names, files, ora, db


def _cmp(name, what, got, exp, stride):
    if not np.array_equal(got, exp):
        bad = np.flatnonzero(got != exp)
        first = int(bad[0])
        raise AssertionError("%s %s: %d bytes differ, first at row %d col %d (got %d, want %d)" %
                             (name, what, bad.size, first // stride, first % stride, got[first], exp[first]))


def test_recon_matches_oracle_unfiltered(dev_batch):
    import torch
    names, files, ora, db = dev_batch
    db.recon()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        y, u, v = db.planes(i)
        _cmp(n, "Y", y, ora[i]["y0"], ora[i]["ystride"])
        _cmp(n, "U", u, ora[i]["u0"], ora[i]["uvstride"])
        _cmp(n, "V", v, ora[i]["v0"], ora[i]["uvstride"])


def test_filter_matches_oracle(dev_batch):
    import torch
    names, files, ora, db = dev_batch
    db.recon()
    db.loopfilter()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        y, u, v = db.planes(i)
        _cmp(n, "Y", y, ora[i]["y"], ora[i]["ystride"])
        _cmp(n, "U", u, ora[i]["u"], ora[i]["uvstride"])
        _cmp(n, "V", v, ora[i]["v"], ora[i]["uvstride"])


def test_decode_to_argb_matches_golden(dev_batch):
    import torch
    names, files, ora, db = dev_batch
    db.decode_to_argb()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        got = db.argb_of(i)
        if n in LOSSY:
            exp = golden_expected(n)[0].reshape(-1)
            _cmp(n, "ARGB vs reference golden", got, exp, 4 * ora[i]["width"])
        else:
            # a.webp has no golden; alpha is handled in test_gpu_api.py and test_gpu_vp8l.py, here the frame is converted opaque
            rgb_exp = ora[i]["to_argb"].reshape(-1, 4)[:, 1:]
            assert np.array_equal(got.reshape(-1, 4)[:, 1:], rgb_exp)


def test_decode_to_argb_also_returns_filtered_planes(dev_batch):
    import torch
    names, files, ora, db = dev_batch
    db.y.zero_(); db.u.zero_(); db.v.zero_()
    db.decode_to_argb(planes=True)
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        y, u, v = db.planes(i)
        _cmp(n, "Y", y, ora[i]["y"], ora[i]["ystride"])
        _cmp(n, "U", u, ora[i]["u"], ora[i]["uvstride"])
        _cmp(n, "V", v, ora[i]["v"], ora[i]["uvstride"])


def test_staged_workspace_path_equals_fused(dev_batch):
    """recon -> loop filter -> finishing pass on the tiled workspace only (what bench.py times per stage)."""
    import torch
    names, files, ora, db = dev_batch
    db.decode_to_argb(planes=False)
    torch.cuda.synchronize()
    ref = db.argb.clone()
    db.argb.zero_()
    db.recon(planes=False)
    db.loopfilter(planes=False)
    db.finish(planes=False, argb=True)
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        d = db.b.desc[i]
        got, exp = db.argb[d.argb_off:d.argb_off + 4 * d.width * d.height], ref[d.argb_off:d.argb_off + 4 * d.width * d.height]
        if not torch.equal(got, exp):
            bad = torch.nonzero(got != exp).flatten()
            raise AssertionError("staged != fused for %s: %d bytes differ, first at pixel (%d, %d)" % (
                n, bad.numel(), int(bad[0]) // 4 % d.width, int(bad[0]) // 4 // d.width))
    # the row-major conversion entry point (Tools.toArgb on given planes) agrees with the finishing pass
    db.decode_to_argb(planes=True)
    db.argb.zero_()
    db.convert()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        d = db.b.desc[i]
        assert torch.equal(db.argb[d.argb_off:d.argb_off + 4 * d.width * d.height], ref[d.argb_off:d.argb_off + 4 * d.width * d.height]), n


def test_one_launch_recon_filter_equals_two_launches(dev_batch):
    """wp_vp8_recon_filter_dev (each workgroup reconstructs its frames, then filters them) leaves the workspace as
    wp_vp8_recon_dev + wp_vp8_loopfilter_dev do: same planes and ARGB from the finishing pass, also through
    LossyPipeline(one_launch=True)."""
    import torch
    from webp_amd.device import LossyPipeline
    names, files, ora, db = dev_batch
    db.recon(planes=False); db.loopfilter(planes=False); db.finish()
    torch.cuda.synchronize()
    ref = [t.clone() for t in (db.argb, db.y, db.u, db.v)]
    for variant in ("direct", "pipeline"):
        for t in (db.argb, db.y, db.u, db.v):
            t.zero_()
        db.work.zero_()
        if variant == "direct":
            db.recon_filter(); db.finish()
        else:
            pipe = LossyPipeline(one_launch=True)
            for _ in range(3):
                pipe.submit(db)
            pipe.drain()
        torch.cuda.synchronize()
        for i in range(db.b.n):
            d = db.b.desc[i]
            nmb = d.mbw * d.mbh
            for got, exp, off, sz, what in ((db.argb, ref[0], d.argb_off, 4 * d.width * d.height, "argb"), (db.y, ref[1], d.y_off, 256 * nmb, "y"),
                                            (db.u, ref[2], d.u_off, 64 * nmb, "u"), (db.v, ref[3], d.v_off, 64 * nmb, "v")):
                assert torch.equal(got[off:off + sz], exp[off:off + sz]), (variant, names[i] if i < len(names) else i, what)


@pytest.mark.parametrize("split", [0.5, 0.0, 0.3, 1.0])
def test_pipelined_steps_equal_serial(dev_batch, split):
    """LossyPipeline (the finishing pass of one step, in two frame ranges, on a second stream beside reconstruction and
    loop filter of the next step; alternating workspaces) gives the same bytes as the serial path, for the same batch
    resubmitted and for two batches alternating."""
    import glob
    import os
    import torch
    from conftest import ROOT
    from webp_amd import parse
    from webp_amd.decoder import LossyBatch
    from webp_amd.device import DeviceLossyBatch, LossyPipeline
    names, files, ora, db = dev_batch
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy1080_q*_[0-2].webp")))
    assert paths
    other = [parse(open(p, "rb").read()) for p in paths]
    db2 = DeviceLossyBatch(LossyBatch([other[i % len(other)].frames[0] for i in range(24)]))
    refs = []
    for d in (db, db2):
        d.decode_to_argb()
        torch.cuda.synchronize()
        refs.append([t.clone() for t in (d.argb, d.y, d.u, d.v)])
        for t in (d.argb, d.y, d.u, d.v):
            t.zero_()
    pipe = LossyPipeline(split=split)
    for k in range(7):
        pipe.submit((db, db2)[k & 1] if k < 4 else db2)
    pipe.drain()
    torch.cuda.synchronize()
    for d, ref in zip((db, db2), refs):
        for got, exp, what in zip((d.argb, d.y, d.u, d.v), ref, ("argb", "y", "u", "v")):
            assert torch.equal(got, exp), what


def test_workspace_checks():
    """A workspace that is too small, or descriptors that were not planned, are refused before any launch."""
    import ctypes
    import torch
    from webp_amd import parse
    from webp_amd._lib import WebPError, check, lib
    from webp_amd.decoder import LossyBatch
    from webp_amd.device import DeviceLossyBatch
    db = DeviceLossyBatch(LossyBatch([parse(golden_input("1")).frames[0]] * 2))
    with pytest.raises(WebPError, match="workspace too small"):
        check(lib().wp_vp8_recon_dev(db.b.n, db.b.desc, db.frames.data_ptr(), db.mb.data_ptr(), db.coeff.data_ptr(),
                                     db.work.data_ptr(), db.work_bytes - 1, None, None, None, None))
    with pytest.raises(WebPError, match="workspace too small"):
        check(lib().wp_vp8_recon_filter_dev(db.b.n, db.b.desc, db.frames.data_ptr(), db.mb.data_ptr(), db.filt.data_ptr(), db.coeff.data_ptr(),
                                            db.work.data_ptr(), db.work_bytes - 1, None))
    with pytest.raises(WebPError, match="null device pointer"):
        check(lib().wp_vp8_recon_filter_dev(db.b.n, db.b.desc, db.frames.data_ptr(), db.mb.data_ptr(), None, db.coeff.data_ptr(),
                                            db.work.data_ptr(), db.work_bytes, None))
    with pytest.raises(WebPError, match="frame range outside the batch"):
        check(lib().wp_vp8_finish_range_dev(db.b.n, db.b.desc, db.frames.data_ptr(), 1, 2, db.work.data_ptr(), db.work_bytes,
                                            None, None, None, None, db.argb.data_ptr(), None))
    saved = db.b.desc[1].stile_base
    db.b.desc[1].stile_base = 0
    with pytest.raises(WebPError, match="wp_vp8_plan"):
        db.recon(planes=False)
    db.b.desc[1].stile_base = saved
    db.decode_to_argb()
    torch.cuda.synchronize()
