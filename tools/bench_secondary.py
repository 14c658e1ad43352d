"""The other BASELINE.json configurations, timed inside the driver-captured bench.py line (`"secondary": [...]`):
  cfg 3  1024 x 1920x1080 VP8L lossless: inverse transform chain + RGBA->ARGB on device-resident decodePix output
  cfg 3b the same on photo-like streams whose tiles mix predictor modes (not a BASELINE configuration: cfg 3's noise images
         use one mode throughout, which is the predictor pass's best case)
  cfg 4  animated WebP, 300 x 1280x720 lossy frames with VP8L-coded ALPH planes (a real ANMF file, tools/make_anim.py)
  cfg 5  this GPU's share of 8192 x 3840x2160 lossy over 8 GPUs: 1024 x 4K, the same pipeline as the headline
  e2e    1024 x 1080p lossy, file bytes in host memory -> ARGB + planes in pinned host memory (LossyStream)
Every record: ms per batch, Mpix/s, algorithmic bytes, fraction of the 8 TB/s HBM peak, and the number of
frames compared bit for bit with the oracle after the timed region.  Only bench.py (rank 0, one GPU) calls this."""
import ctypes
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
HBM_PEAK = 8000.0e9


def _timed(fn, reps):
    import torch
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sum(ts) / len(ts)


def _record(name, workload, ms, px, alg_bytes, verified, **extra):
    r = {"name": name, "workload": workload, "ms": round(ms, 4), "Mpix_s": round(px / ms / 1e3, 1), "alg_bytes": int(alg_bytes),
         "frac": round(alg_bytes / (ms * 1e-3) / HBM_PEAK, 4), "verified_frames": verified}
    r.update(extra)
    return r


def lossless(device, n=1024, reps=4, pattern="lossless1080_*.webp", name="cfg3_lossless_1080p",
             what="predictor + cross-colour + subtract-green, and a palette image"):
    import numpy as np
    import torch
    from oracle import oracle
    from webp_amd import parse
    from webp_amd._lib import check, lib
    from webp_amd.decoder import LosslessBatch
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", pattern)))
    datas = [open(p, "rb").read() for p in paths]
    files = [parse(d) for d in datas]
    b = LosslessBatch([files[i % len(files)].frames[0] for i in range(n)])
    d_frames = torch.from_numpy(np.frombuffer(bytes(b.desc), dtype=np.uint8).copy()).to(device)
    pix0 = torch.from_numpy(b.pix).to(device)
    pix = torch.empty_like(pix0)
    aux = torch.from_numpy(b.aux).to(device)
    tmp = torch.empty(b.out_bytes, dtype=torch.uint8, device=device)
    out = torch.empty(b.out_bytes, dtype=torch.uint8, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    ts = []
    for it in range(reps + 1):
        pix.copy_(pix0)                 # the chain consumes its input (transforms run in place, as in the reference): restored outside the timed span
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().wp_vp8l_inverse_dev(n, b.desc, d_frames.data_ptr(), pix.data_ptr(), aux.data_ptr(), tmp.data_ptr(), out.data_ptr(), 1, stream))
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sum(ts[1:]) / reps
    want = [oracle.decode(d)["argb"] for d in datas]
    idx = sorted(set([0, 1, 2, n // 2, n - 3, n - 2, n - 1]))
    for i in idx:
        d = b.desc[i]
        got = out[d.out_off:d.out_off + 4 * d.width * d.height].cpu().numpy()
        if not np.array_equal(got, want[i % len(files)]):
            raise RuntimeError("lossless frame %d differs from the oracle" % i)
    px = sum(b.desc[i].width * b.desc[i].height for i in range(n))
    return _record(name, "%d x 1920x1080 VP8L (%d libwebp streams cycled: %s): inverse chain + ARGB"
                   % (n, len(files), what), ms, px, b.pix.size + b.aux.size + 4 * px, len(idx))


def lossless_mixed(device):
    """cfg 3's shape on photo-like streams (tools/make_mixed_lossless.py): libwebp mixes predictor modes tile by tile there
    (select and clamp-add-subtract; or mode 10 with a sprinkling of 7 and 8), so the wavefronts of the predictor pass hold several
    modes at once."""
    return lossless(device, pattern="ll1080_*_m*.webp", name="cfg3b_lossless_1080p_mixed_modes",
                    what="photo-like image at methods 4 and 6: predictor with select / clamp-add-subtract mixed tile by tile; smooth image: "
                         "mode 10 with a few tiles of 7 and 8; all with cross-colour")


def anim_alpha(device, n=300, reps=4):
    import numpy as np
    import torch
    from make_anim import build_anim
    from oracle import oracle
    from webp_amd import parse
    from webp_amd._lib import AlphaFrame, check, lib
    from webp_amd.decoder import LossyBatch, LosslessBatch, _load_vp8l_meta
    from webp_amd.device import DeviceLossyBatch
    stills = [open(p, "rb").read() for p in sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy720_alpha_*.webp")))]
    anim = build_anim(stills, n)
    t0 = time.perf_counter()
    pf = parse(anim)                                           # one ANMF file: container walk + entropy decode of its 300 frames
    t_parse = time.perf_counter() - t0
    frames = pf.frames
    assert pf.kind == 2 and len(frames) == n
    for f in frames:
        _load_vp8l_meta(f)
        assert f.has_alpha and f.alpha_is_vp8l
    # One round by default: the whole animation as ONE device batch through the library's pipeline (wp_vp8_pipeline_*).  The
    # alpha planes (alpha chain with the green-only last pass writing the alpha slab + unfilterAlpha in place) are needed by
    # the finishing pass only: they are produced on a side stream from the start and the finishing pass waits for their event.
    # What bounds the step is the latency of ONE frame: a 720p frame is a chain of 174 dependent wavefront steps and takes
    # 0.90 ms with the GPU to itself, 0.94 ms when 256 frames run at once (one per CU) and 1.30 ms for 300 (two workgroups on
    # 44 CUs) -- tools/mb_small_batches.py.  Rounds of one frame per CU, the finishing pass of a round beside the kernel of
    # the next (WEBP_CFG4_ROUND=256), do not pay: 0.94 + 0.91 ms of kernels in sequence, 2.70 ms per step against 1.9.
    from webp_amd.device import NativePipeline, pipeline_streams
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    per_round = int(os.environ.get("WEBP_CFG4_ROUND", "0")) or n
    bounds = list(range(0, n, per_round)) + [n]
    parts = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        fr = frames[lo:hi]
        m = len(fr)
        batch = LossyBatch(fr, with_alpha=[True] * m)
        lb = LosslessBatch(fr, alpha_image=True)
        for i in range(m):
            lb.desc[i].out_off = batch.desc[i].alpha_off      # the chain's green-only last pass writes the alpha slab itself
        db = DeviceLossyBatch(batch, device, alpha=np.zeros(max(batch.alpha_bytes, 16), np.uint8))
        adesc = (AlphaFrame * m)()                             # unfilterAlpha in place on the alpha slab
        for i, f in enumerate(fr):
            a = adesc[i]
            a.width, a.height, a.filter, a.src_is_vp8l_rgba = f.alpha_width, f.alpha_height, f.alpha_filter, 0
            a.src_off = a.dst_off = batch.desc[i].alpha_off
        pix0 = torch.from_numpy(lb.pix).to(device)
        parts.append(dict(lo=lo, m=m, batch=batch, lb=lb, db=db, adesc=adesc,
                          d_lf=torch.from_numpy(np.frombuffer(bytes(lb.desc), dtype=np.uint8).copy()).to(device),
                          d_af=torch.from_numpy(np.frombuffer(bytes(adesc), dtype=np.uint8).copy()).to(device),
                          pix0=pix0, pix=torch.empty_like(pix0), aux=torch.from_numpy(lb.aux).to(device),
                          tmp=torch.empty(lb.out_bytes, dtype=torch.uint8, device=device),
                          unfilter=any(f.alpha_filter != 0 for f in fr)))
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    s_side = pipeline_streams(device)[1]
    side = ctypes.c_void_p(s_side.cuda_stream)
    pipe = NativePipeline(device, work_bytes=max(p_["db"].work_bytes for p_ in parts), workspaces=2)

    def step():
        s_main = torch.cuda.current_stream(device)
        ev0 = torch.cuda.Event(); ev0.record(s_main)
        ready = []
        with torch.cuda.stream(s_side):
            s_side.wait_event(ev0)
            for p_ in parts:
                p_["pix"].copy_(p_["pix0"])                    # the chain consumes its input (1.2 B/px of the frame: inside the timed span here)
                check(lib().wp_vp8l_inverse_dev(p_["m"], p_["lb"].desc, p_["d_lf"].data_ptr(), p_["pix"].data_ptr(), p_["aux"].data_ptr(),
                                                p_["tmp"].data_ptr(), p_["db"].alpha.data_ptr(), 2, side))
                if p_["unfilter"]:
                    check(lib().wp_alpha_unfilter_dev(p_["m"], p_["adesc"], p_["d_af"].data_ptr(), p_["db"].alpha.data_ptr(), p_["db"].alpha.data_ptr(), side))
                ev = torch.cuda.Event(); ev.record(s_side)
                ready.append(ev)
        for p_, ev in zip(parts, ready):
            pipe.submit(p_["db"], alpha_ready=ev)
        pipe.flush()                                           # the current stream waits for everything

    ms = _timed(step, reps)
    pipe.drain()
    want = oracle.decode(anim)["frames"]
    idx = sorted(set([0, 1, 2, n // 2, min(cus, n - 1), n - 2, n - 1]))
    def part_of(i):
        for p_ in parts:
            if p_["lo"] <= i < p_["lo"] + p_["m"]:
                return p_["db"], i - p_["lo"]
    for i in idx:
        dbi, k = part_of(i)
        if not np.array_equal(dbi.argb_of(k), want[i]["to_argb"]):
            raise RuntimeError("frame %d of the animation (A,R,G,B with its decoded alpha) differs from the oracle" % i)
        y, u, v = dbi.planes(k)
        if not (np.array_equal(y, want[i]["y"]) and np.array_equal(u, want[i]["u"]) and np.array_equal(v, want[i]["v"])):
            raise RuntimeError("planes of frame %d of the animation differ from the oracle" % i)
    pipe.close()
    px = sum(p_["batch"].desc[i].width * p_["batch"].desc[i].height for p_ in parts for i in range(p_["m"]))
    alg = sum(p_["batch"].input_bytes() + 384 * p_["batch"].n_mb + p_["lb"].pix.size + p_["lb"].aux.size for p_ in parts) + 4 * px + px        # + alpha plane read, + the coded alpha image
    # canvas compositing wants the frames of the whole animation in one slab: gather the rounds' ARGB slabs
    argb_all = torch.cat([p_["db"].argb[:p_["batch"].argb_bytes] for p_ in parts])
    argb_off = []
    base = 0
    for p_ in parts:
        argb_off += [base + int(p_["batch"].desc[i].argb_off) for i in range(p_["m"])]
        base += p_["batch"].argb_bytes
    # canvas compositing of the same animation on the device: frames are where the finishing pass left them
    from webp_amd._lib import AnimFrame
    cdesc = (AnimFrame * n)()
    for i, f in enumerate(frames):
        c = cdesc[i]
        c.x, c.y, c.width, c.height, c.blend, c.dispose, c.argb_off = f.x, f.y, f.width, f.height, int(f.blend), int(f.dispose), argb_off[i]
    d_cf = torch.from_numpy(np.frombuffer(bytes(cdesc), dtype=np.uint8).copy()).to(device)
    canvases = torch.empty(n * 4 * pf.width * pf.height, dtype=torch.uint8, device=device)
    compose = lambda: check(lib().wp_anim_compose_dev(n, cdesc, d_cf.data_ptr(), pf.width, pf.height, argb_all.data_ptr(), canvases.data_ptr(), stream))
    compose_ms = _timed(compose, reps)
    k_chk = 4
    want_c = oracle.anim_compose([(w["x"], w["y_off"], w["width"], w["height"], w["blend"], w["dispose"], w["to_argb"]) for w in want[:k_chk]], pf.width, pf.height)
    got_c = canvases[:k_chk * 4 * pf.width * pf.height].cpu().numpy().reshape(want_c.shape)
    if not np.array_equal(got_c, want_c):
        raise RuntimeError("composed canvases of the animation differ from the oracle")
    compose_bytes = 4 * px + 4 * n * pf.width * pf.height
    extra_compose = {"compose_ms": round(compose_ms, 4), "compose_Mpix_s": round(n * pf.width * pf.height / compose_ms / 1e3, 1),
                     "compose_frac": round(compose_bytes / (compose_ms * 1e-3) / 8e12, 4), "compose_verified_canvases": k_chk}
    return _record("cfg4_anim_720p_alpha", "one animated WebP (ANMF), %d frames of 1280x720 lossy + VP8L-coded ALPH (filter %d): one device batch through wp_vp8_pipeline_*; alpha chain + unfilterAlpha on a side stream beside the reconstruction + loop filter kernel, awaited by the finishing pass only"
                   % (n, frames[0].alpha_filter), ms, px, alg, len(idx), host_parse_s=round(t_parse, 3), file_bytes=len(anim), **extra_compose)


def vp8l_resolve(device, n=2048):
    """decodePix's pixel loop on the device (first slice): n copies of a photo-like lossless stream's tokens."""
    import numpy as np
    import torch
    from oracle import oracle
    from webp_amd import parse
    from webp_amd._lib import Vp8lTokensFrame, check, lib
    path = os.path.join(ROOT, "tests", "golden", "inputs", "3ll.webp")
    f = parse(open(path, "rb").read(), vp8l_tokens=True).frames[0]
    tk = f.buf(6, np.uint32)
    w, h = f.coded_width, f.vp8l_height
    stride = (4 * w * h + 255) // 256 * 256
    desc = (Vp8lTokensFrame * n)()
    for i in range(n):
        d = desc[i]
        d.coded_width, d.height, d.cc_bits, d.tok_off, d.n_words, d.pix_off = w, h, f.cc_bits, 0, tk.size, stride * i
    tok = torch.from_numpy(tk.view(np.int32).copy()).to(device)
    dd = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(device)
    pix = torch.empty(stride * n, dtype=torch.uint8, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    ms = _timed(lambda: check(lib().wp_vp8l_resolve_dev(n, desc, dd.data_ptr(), tok.data_ptr(), pix.data_ptr(), stream)), 3)
    want = oracle.resolve_tokens(w, h, f.cc_bits, tk)
    idx = [0, n // 2, n - 1]
    for i in idx:
        if not np.array_equal(pix[stride * i:stride * i + 4 * w * h].cpu().numpy(), want):
            raise RuntimeError("resolved pixels of image %d differ from the oracle" % i)
    return _record("f2_vp8l_pixel_loop", "%d x %dx%d VP8L token streams (photo-like, %d tokens words each, colour cache of %d entries): literal runs + LZ77 copies + "
                   "colour-cache fill / lookup (Vp8LDecoder.hx:283-333) on the device, one wavefront per image" % (n, w, h, tk.size, 1 << f.cc_bits),
                   ms, n * w * h, n * (4 * tk.size + 4 * w * h), len(idx), note="latency-bound per token; the host's Huffman decode still precedes it")


def lossy_4k(device, n=1024, steps=16):
    import numpy as np
    import torch
    import bench
    from oracle import oracle
    from webp_amd.device import LossyPipeline
    files, batch, db, paths, _ = bench.build_batch(n, "lossy2160_q75_0.webp", device)
    pipe = LossyPipeline(device, one_launch=True)
    db.workspace(1)
    for _ in range(2):
        pipe.submit(db)
    pipe.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(db)
    pipe.drain(); torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    want = oracle.decode(open(paths[0], "rb").read())
    idx = sorted(set([0, n // 2 + 1, n - 1]))
    for i in idx:
        y, u, v = db.planes(i)
        if not (np.array_equal(y, want["y"]) and np.array_equal(u, want["u"]) and np.array_equal(v, want["v"]) and np.array_equal(db.argb_of(i), want["to_argb"])):
            raise RuntimeError("4K frame %d differs from the oracle" % i)
    px = sum(batch.desc[i].width * batch.desc[i].height for i in range(n))
    alg = batch.input_bytes() + 384 * batch.n_mb + 4 * px
    return _record("cfg5_lossy_4k_per_gpu", "%d x 3840x2160 VP8 lossy (one GPU's share of 8192 over 8 GPUs; libwebp q75, normal filter): recon + loop filter + finishing pass, steps pipelined as in the headline"
                   % n, ms, px, alg, len(idx))


def end_to_end(device, n=1024, threads=16, chunk=32):
    import numpy as np
    import torch
    from oracle import oracle
    from webp_amd.streaming import LossyStream
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy1080_*.webp")))
    uniq = [open(p, "rb").read() for p in paths]
    datas = [uniq[i % len(uniq)] for i in range(n)]
    stream = LossyStream(device, threads=threads, chunk=chunk, planes=True)
    stream.decode(datas)                                       # warm-up: streams, pinned staging and output slabs (a serving loop keeps them)
    torch.cuda.synchronize()
    dt = 1e9
    for _ in range(2):                                         # host-side timing on a shared machine: the better of two runs
        t0 = time.perf_counter()
        argb, items, y, u, v = stream.decode(datas)
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    idx = sorted(set([0, chunk - 1, chunk, n // 2, n - 1]))
    for i in idx:
        want = oracle.decode(datas[i])
        it = items[i]
        if not np.array_equal(argb[it["argb_off"]:it["argb_off"] + want["to_argb"].size].numpy(), want["to_argb"]):
            raise RuntimeError("streamed image %d differs from the oracle" % i)
    px = sum(it["width"] * it["height"] for it in items)
    file_bytes = sum(len(d) for d in datas)
    r = _record("e2e_lossy_1080p_stream", "%d x 1920x1080 lossy files in host memory -> ARGB + YUV420 planes in pinned host memory: %d host threads parse | pinned gather + H2D | kernels | D2H, chunks of %d"
                % (n, threads, chunk), 1e3 * dt, px, file_bytes + 5.5 * px, len(idx), host_threads=threads)
    r["note"] = "PCIe- and host-parse-inclusive: bound by the host entropy decode, not by HBM; never the headline value"
    return r


def run_all(device):
    import gc
    import torch
    out = []
    for fn in (lossless, lossless_mixed, anim_alpha, lossy_4k, vp8l_resolve, end_to_end):
        t0 = time.time()
        try:
            rec = fn(device)
        except Exception as e:                                  # a secondary record never takes the headline line down
            rec = {"name": fn.__name__, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        rec["wall_s"] = round(time.time() - t0, 1)
        out.append(rec)
        gc.collect()
        torch.cuda.empty_cache()
    return out
