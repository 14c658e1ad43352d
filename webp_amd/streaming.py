"""End-to-end decode of many lossy stills (with or without an ALPH chunk): file bytes in host memory -> ARGB (and, when
asked, YUV420 planes) in pinned host memory, in chunks that move through four overlapping stages:

    host threads: container walk + entropy decode of the files ahead (wp_parse_job_*: one work queue for all files)
    this thread : batch assembly of chunk k into pinned slabs (wp_vp8_batch_layout / _gather), async H2D
    GPU         : reconstruction, loop filter, finishing pass of chunk k (compute stream)
    GPU DMA     : D2H of chunk k-1's outputs (copy stream)

Two slots of device / pinned buffers alternate; events order the stages, nothing blocks on the GPU except a slot that is
still in flight when it comes round again.  This is SURVEY 8(f) rank 1 (host parse threading + H2D), kept out of
bench.py's `value`, which is the device-resident rate; tools/e2e_bench.py times it."""
import ctypes
from typing import List

import numpy as np
import torch

from ._lib import AlphaFrame, Vp8Frame, Vp8lFrame, check, lib
from .decoder import LossyBatch, LosslessBatch, ParseJob, _load_vp8l_meta

_STREAMS = {}


def stream_streams(device):
    """The three streams (upload, compute, download) every LossyStream of the process shares on a device: the number of
    streams the library creates per process stays fixed, whatever the number of objects."""
    device = torch.device(device)
    key = device.index or 0
    if key not in _STREAMS:
        _STREAMS[key] = tuple(torch.cuda.Stream(device) for _ in range(3))
    return _STREAMS[key]


class _Slot:
    """Grow-only pinned and device buffers of one pipeline slot."""

    def __init__(self, device):
        self.device = device
        self.pin = {}
        self.dev = {}
        self.done = None            # event: the slot's D2H copies have finished

    def pinned(self, name, nbytes):
        t = self.pin.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes * 1.25), 256), dtype=torch.uint8, pin_memory=True)
            self.pin[name] = t
        return t

    def on_device(self, name, nbytes):
        t = self.dev.get(name)
        if t is None or t.numel() < nbytes:
            t = torch.empty(max(int(nbytes * 1.25), 256), dtype=torch.uint8, device=self.device)
            self.dev[name] = t
        return t


class LossyStream:
    def __init__(self, device="cuda:0", threads=16, chunk=32, planes=True):
        self.device = torch.device(device)
        self.threads, self.chunk, self.planes = threads, chunk, planes
        lib().wp_init(self.device.index or 0)
        self.s_in, self.s_compute, self.s_out = stream_streams(self.device)
        self.slots = [_Slot(self.device), _Slot(self.device)]
        self._out_cache = {}            # pinned output slabs of the previous decode(), reused when large enough (allocating ~13 GB of pinned memory costs ~0.4 s)

    def decode(self, datas: List[bytes]):
        """Returns (argb, items[, y, u, v]): pinned uint8 tensors holding every image's output and, per image, a dict with
        width, height, mbw, mbh and the byte offsets of its outputs.  A still with an ALPH chunk gets its alpha plane decoded
        on the device (VP8L chain with the green-only last pass, unfilterAlpha: WebPDecoder.hx:311-411): it is byte 0 of every
        A,R,G,B pixel, and item["alpha"] holds the plane itself (what ImageData.Yuv420.a carries) as a numpy array.  The
        output tensors belong to this object and are reused by its next decode() (copy what must outlive it).  Raises
        WebPError on the first file that does not parse, ValueError on one that is not a lossy still."""
        n = len(datas)
        chunks = [datas[i:i + self.chunk] for i in range(0, n, self.chunk)]
        job = ParseJob(datas, self.threads, window=3 * self.chunk)     # one work queue; chunks are taken in index order
        items = []
        outs = {}                               # name -> [pinned tensor, bytes used]

        def out_slab(name, need):
            t, used = outs.get(name, (None, 0))
            if t is None:                       # sized from the first chunk, with headroom; grows by reallocation if a later chunk needs it
                want = int(need * len(chunks) * 1.1) + 4096
                t = self._out_cache.get(name)
                if t is None or t.numel() < want:
                    t = torch.empty(want, dtype=torch.uint8, pin_memory=True)
            elif used + need > t.numel():
                torch.cuda.synchronize(self.device)
                t2 = torch.empty(int((used + need) * 1.5), dtype=torch.uint8, pin_memory=True)
                t2[:used] = t[:used]
                t = t2
            outs[name] = [t, used]
            return t, used

        names = ("argb", "y", "u", "v") if self.planes else ("argb",)
        try:
            return self._decode_chunks(job, chunks, n, names, items, outs, out_slab)
        finally:
            job.close()                         # also on an error: the parser threads stop and the parsed files are released

    def _stage_alpha(self, slot, b, frames, with_alpha):
        """Uploads what the alpha planes of a chunk are made from (on the upload stream) and returns the device alpha slab and
        the descriptors of the device work: raw ALPH payloads go straight to their place in the slab, VP8L-coded ones as the
        packed buffers of wp_vp8l_inverse_dev, whose green-only last pass writes into the slab."""
        if not any(with_alpha):
            return None, None
        idx = [i for i, a in enumerate(with_alpha) if a]
        for i in idx:
            _load_vp8l_meta(frames[i])
        d_alpha = slot.on_device("alpha", max(b.alpha_bytes, 16))
        coded = [i for i in idx if frames[i].alpha_is_vp8l]
        jobs = {"n": len(idx), "lb": None}
        with torch.cuda.stream(self.s_in):
            for i in idx:
                if not frames[i].alpha_is_vp8l:
                    raw = frames[i].buf(4)
                    stage = slot.pinned("alpha_raw_%d" % i, raw.size)
                    stage[:raw.size].numpy()[:] = raw
                    off = int(b.desc[i].alpha_off)
                    d_alpha[off:off + raw.size].copy_(stage[:raw.size], non_blocking=True)
            if coded:
                lb = LosslessBatch([frames[i] for i in coded], alpha_image=True)
                for k, i in enumerate(coded):
                    lb.desc[k].out_off = b.desc[i].alpha_off
                up = {}
                for name, arr in (("a_pix", lb.pix), ("a_aux", lb.aux), ("a_desc", np.frombuffer(bytes(lb.desc), np.uint8)[:ctypes.sizeof(Vp8lFrame) * lb.n])):
                    stage = slot.pinned(name, arr.size)
                    stage[:arr.size].numpy()[:] = arr
                    up[name] = slot.on_device(name, arr.size)
                    up[name][:arr.size].copy_(stage[:arr.size], non_blocking=True)
                jobs["lb"] = (lb, up, slot.on_device("a_tmp", lb.out_bytes))
            adesc = (AlphaFrame * len(idx))()
            for k, i in enumerate(idx):
                a, f = adesc[k], frames[i]
                a.width, a.height, a.filter, a.src_is_vp8l_rgba = f.alpha_width, f.alpha_height, f.alpha_filter, 0
                a.src_off = a.dst_off = b.desc[i].alpha_off                     # unfiltered in place
            raw = np.frombuffer(bytes(adesc), np.uint8)
            stage = slot.pinned("a_adesc", raw.size)
            stage[:raw.size].numpy()[:] = raw
            d_adesc = slot.on_device("a_adesc", raw.size)
            d_adesc[:raw.size].copy_(stage[:raw.size], non_blocking=True)
            jobs["adesc"] = (adesc, d_adesc)
        return d_alpha, jobs

    def _run_alpha(self, jobs, d_alpha):
        """On the compute stream: alpha images -> green bytes in the slab (WebPDecoder.hx:330-349), then unfilterAlpha (:357-411)."""
        stream = ctypes.c_void_p(self.s_compute.cuda_stream)
        if jobs["lb"] is not None:
            lb, up, tmp = jobs["lb"]
            check(lib().wp_vp8l_inverse_dev(lb.n, lb.desc, up["a_desc"].data_ptr(), up["a_pix"].data_ptr(), up["a_aux"].data_ptr(),
                                            tmp.data_ptr(), d_alpha.data_ptr(), 2, stream))
        adesc, d_adesc = jobs["adesc"]
        check(lib().wp_alpha_unfilter_dev(jobs["n"], adesc, d_adesc.data_ptr(), d_alpha.data_ptr(), d_alpha.data_ptr(), stream))

    def _decode_chunks(self, job, chunks, n, names, items, outs, out_slab):
        for k in range(len(chunks)):
            files = job.take(min(n, (k + 1) * self.chunk))
            frames = []
            for f in files:
                if f.kind != 0 or not f.frames or f.frames[0].type != 0:
                    raise ValueError("LossyStream handles lossy still images only")
                frames.append(f.frames[0])
            slot = self.slots[k & 1]
            if slot.done is not None:
                slot.done.synchronize()         # the slot's previous chunk has left the device
            order = iter(("mb", "filt", "coeff"))
            with_alpha = [bool(f.has_alpha) for f in frames]
            b = LossyBatch(frames, with_alpha=with_alpha if any(with_alpha) else None,
                           alloc=lambda nbytes: slot.pinned(next(order), nbytes).numpy(), threads=self.threads)
            desc_bytes = ctypes.sizeof(Vp8Frame) * b.n
            pd = slot.pinned("desc", desc_bytes)
            ctypes.memmove(pd.data_ptr(), ctypes.addressof(b.desc), desc_bytes)
            sizes = {"mb": b.mb.nbytes, "filt": b.filt.nbytes, "coeff": b.coeff.nbytes, "desc": desc_bytes}
            with torch.cuda.stream(self.s_in):
                for name, nb in sizes.items():
                    slot.on_device(name, nb)[:nb].copy_(slot.pin[name][:nb], non_blocking=True)
                ev_in = torch.cuda.Event()
                ev_in.record(self.s_in)
            need = {"argb": b.argb_bytes, "y": b.y_bytes, "u": b.u_bytes, "v": b.v_bytes}
            dev_out = {name: slot.on_device("out_" + name, need[name]) for name in names}
            work = slot.on_device("work", max(b.work_bytes, 256))
            d_alpha, alpha_jobs = self._stage_alpha(slot, b, frames, with_alpha)
            if d_alpha is not None:
                ev_in = torch.cuda.Event()
                ev_in.record(self.s_in)          # the alpha uploads were queued after the first event
            with torch.cuda.stream(self.s_compute):
                self.s_compute.wait_event(ev_in)
                if d_alpha is not None:
                    self._run_alpha(alpha_jobs, d_alpha)
                pl = [dev_out[p].data_ptr() if self.planes else None for p in ("y", "u", "v")]
                check(lib().wp_vp8_decode_to_argb_dev(b.n, b.desc, slot.dev["desc"].data_ptr(), slot.dev["mb"].data_ptr(),
                                                      slot.dev["filt"].data_ptr(), slot.dev["coeff"].data_ptr(), work.data_ptr(),
                                                      b.work_bytes, pl[0], pl[1], pl[2], d_alpha.data_ptr() if d_alpha is not None else None,
                                                      dev_out["argb"].data_ptr(),
                                                      ctypes.c_void_p(self.s_compute.cuda_stream)))
                ev_c = torch.cuda.Event()
                ev_c.record(self.s_compute)
            base = {}
            with torch.cuda.stream(self.s_out):
                self.s_out.wait_event(ev_c)
                for name in names:
                    t, used = out_slab(name, need[name])
                    t[used:used + need[name]].copy_(dev_out[name][:need[name]], non_blocking=True)
                    base[name] = used
                    outs[name][1] = used + need[name]
                host_alpha = None
                if d_alpha is not None:          # the planes themselves (ImageData.Yuv420.a): small, one pinned buffer per slot
                    host_alpha = slot.pinned("alpha_out", b.alpha_bytes)
                    host_alpha[:b.alpha_bytes].copy_(d_alpha[:b.alpha_bytes], non_blocking=True)
                slot.done = torch.cuda.Event()
                slot.done.record(self.s_out)
            if host_alpha is not None:
                slot.done.synchronize()          # chunks with alpha give up the overlap of their download: the planes are copied out now
                host_alpha = host_alpha[:b.alpha_bytes].numpy().copy()
            for i in range(b.n):
                d = b.desc[i]
                it = {"width": d.width, "height": d.height, "mbw": d.mbw, "mbh": d.mbh, "argb_off": base["argb"] + d.argb_off}
                if self.planes:
                    it.update(y_off=base["y"] + d.y_off, u_off=base["u"] + d.u_off, v_off=base["v"] + d.v_off)
                it["alpha"] = host_alpha[d.alpha_off:d.alpha_off + d.width * d.height] if (host_alpha is not None and d.has_alpha) else None
                items.append(it)
            slot.keep = (b, files)              # host objects stay alive until the slot is reused
        for s in self.slots:
            if s.done is not None:
                s.done.synchronize()
        self._out_cache = {name: outs[name][0] for name in outs}      # NOTE: the next decode() on this object overwrites what this one returns
        res = [outs["argb"][0] if outs else torch.empty(0, dtype=torch.uint8), items]
        if self.planes:
            res += [outs[p][0] for p in ("y", "u", "v")] if outs else [torch.empty(0, dtype=torch.uint8)] * 3
        return tuple(res)
