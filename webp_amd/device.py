"""Device-resident batches: packed slabs held in HBM (torch is used only as the allocator / stream
provider) and thin wrappers over the *_dev entry points of the C ABI."""
import ctypes

import numpy as np
import torch

from ._lib import Vp8Frame, check, lib
from .decoder import LossyBatch


def _dev(a: np.ndarray, device):
    if a.size == 0:
        a = np.zeros(16, a.dtype)
    t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy())
    return t.to(device, non_blocking=False)


class DeviceLossyBatch:
    """A LossyBatch uploaded once; recon / filter / convert run on it with no host traffic."""

    def __init__(self, batch: LossyBatch, device="cuda:0", alpha: np.ndarray = None):
        self.b = batch
        self.device = torch.device(device)
        lib().wp_init(self.device.index or 0)
        work_bytes = ctypes.c_uint64(0)
        check(lib().wp_vp8_plan(batch.n, batch.desc, ctypes.byref(work_bytes)))      # fills stile_base before the upload
        self.work_bytes = int(work_bytes.value)
        self.work = torch.empty(max(self.work_bytes, 256), dtype=torch.uint8, device=self.device)   # torch allocations are >= 256-B aligned
        self.frames = _dev(np.frombuffer(bytes(batch.desc), dtype=np.uint8)[:ctypes.sizeof(Vp8Frame) * max(batch.n, 1)], self.device)
        self.mb = _dev(batch.mb, self.device)
        self.filt = _dev(batch.filt, self.device)
        self.coeff = _dev(batch.coeff, self.device)
        self.y = torch.empty(max(batch.y_bytes, 16), dtype=torch.uint8, device=self.device)
        self.u = torch.empty(max(batch.u_bytes, 16), dtype=torch.uint8, device=self.device)
        self.v = torch.empty(max(batch.v_bytes, 16), dtype=torch.uint8, device=self.device)
        self.argb = torch.empty(max(batch.argb_bytes, 16), dtype=torch.uint8, device=self.device)
        self.alpha = _dev(alpha, self.device) if alpha is not None else None
        self.has_unfiltered = any(batch.desc[i].filter_type == 0 for i in range(batch.n))   # read from the R tiles by the finishing pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _planes(self, on):
        return (self.y.data_ptr(), self.u.data_ptr(), self.v.data_ptr()) if on else (None, None, None)

    def _alpha(self):
        return self.alpha.data_ptr() if self.alpha is not None else None

    def recon(self, planes=True):
        """Reconstruction into the tiled workspace; planes=True also writes the unfiltered row-major planes."""
        check(lib().wp_vp8_recon_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.mb.data_ptr(), self.coeff.data_ptr(),
                                     self.work.data_ptr(), self.work_bytes, *self._planes(planes), self._stream()))

    def loopfilter(self, planes=True):
        """planes=True: filter the row-major planes in place; planes=False: workspace R tiles -> S tiles."""
        check(lib().wp_vp8_loopfilter_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.filt.data_ptr(),
                                          self.work.data_ptr(), self.work_bytes, *self._planes(planes), self._stream()))

    def finish(self, planes=True, argb=True):
        check(lib().wp_vp8_finish_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.work.data_ptr(), self.work_bytes,
                                      *self._planes(planes), self._alpha(), self.argb.data_ptr() if argb else None, self._stream()))

    def convert(self):
        check(lib().wp_yuv420_to_argb_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.y.data_ptr(), self.u.data_ptr(),
                                          self.v.data_ptr(), self._alpha(), self.argb.data_ptr(), self._stream()))

    def decode_to_argb(self, planes=True):
        check(lib().wp_vp8_decode_to_argb_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.mb.data_ptr(), self.filt.data_ptr(),
                                              self.coeff.data_ptr(), self.work.data_ptr(), self.work_bytes, *self._planes(planes),
                                              self._alpha(), self.argb.data_ptr(), self._stream()))

    # ---- readback helpers (tests) ----
    def planes(self, i):
        d = self.b.desc[i]
        nmb = d.mbw * d.mbh
        y = self.y[d.y_off:d.y_off + 256 * nmb].cpu().numpy()
        u = self.u[d.u_off:d.u_off + 64 * nmb].cpu().numpy()
        v = self.v[d.v_off:d.v_off + 64 * nmb].cpu().numpy()
        return y, u, v

    def argb_of(self, i):
        d = self.b.desc[i]
        return self.argb[d.argb_off:d.argb_off + 4 * d.width * d.height].cpu().numpy()


class LossyPipeline:
    """Software pipeline over successive device batches: reconstruction and loop filter are enqueued on one HIP stream,
    the finishing pass on a second one, so the memory-bound finishing pass of step k runs while the issue-bound
    reconstruction of step k+1 has already started (measured on 1024 x 1080p: 9.4 -> 8.2 ms per step).

    Hazards on the shared workspace, kept by events: the loop filter of step k+1 rewrites the S tiles the finishing pass
    of step k reads, so it waits for that pass; frames without loop filter are read by the finishing pass from the R
    tiles, so for a batch holding such frames the next reconstruction waits as well (no overlap then).  Outputs (planes,
    ARGB) of a batch are complete once `drain()` has returned or the event `submit()` returns has been waited for.

    Both streams have the default priority, on purpose: with a high-priority finishing stream the pipeline was seen to
    hang once many streams existed in the process (a queue whose barrier packet waits for the other stream's kernel can
    starve that stream when the two share a hardware pipe).  At most `max_in_flight` steps are queued ahead of the GPU."""

    def __init__(self, device="cuda:0", max_in_flight=3):
        self.device = torch.device(device)
        self.s_main = torch.cuda.Stream(self.device)
        self.s_finish = torch.cuda.Stream(self.device)
        self.max_in_flight = max_in_flight
        self._queued = []               # events of the steps not yet known to be complete, oldest first
        self._prev_finish = None        # event of the most recent finishing pass
        self._last_finish = {}          # id(batch) -> event of its last finishing pass
        cur = torch.cuda.current_stream(self.device)
        self.s_main.wait_stream(cur)
        self.s_finish.wait_stream(cur)

    def submit(self, db: "DeviceLossyBatch", planes=True, argb=True, timing=None):
        """Enqueues one step.  `timing`: a list that receives (recon start, recon end, filter start, filter end, finish
        start, finish end) timing events recorded on the stream each kernel is launched on."""
        def mark(stream):
            if timing is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            return e
        while len(self._queued) >= self.max_in_flight:
            self._queued.pop(0).synchronize()
        fin = self._last_finish.get(id(db))
        unfiltered = db.has_unfiltered
        with torch.cuda.stream(self.s_main):
            if fin is not None and unfiltered:
                self.s_main.wait_event(fin)
            r0 = mark(self.s_main)
            db.recon(planes=False)
            r1 = mark(self.s_main)
            # The loop filter waits for the previous finishing pass whichever batch that belonged to: on the same batch it
            # rewrites the S tiles that pass reads, and in any case a persistent kernel that starts while the streaming
            # pass holds the GPU gets its workgroups placed piecemeal and runs far slower (tools/overlap_delay.py).
            if self._prev_finish is not None:
                self.s_main.wait_event(self._prev_finish)
            f0 = mark(self.s_main)
            db.loopfilter(planes=False)
            f1 = mark(self.s_main)
            filtered = torch.cuda.Event()
            filtered.record(self.s_main)
        with torch.cuda.stream(self.s_finish):
            self.s_finish.wait_event(filtered)
            c0 = mark(self.s_finish)
            db.finish(planes, argb)
            c1 = mark(self.s_finish)
            done = torch.cuda.Event()
            done.record(self.s_finish)
        if timing is not None:
            timing.append((r0, r1, f0, f1, c0, c1))
        self._last_finish[id(db)] = done
        self._prev_finish = done
        self._queued.append(done)
        return done

    def drain(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.s_main)
        cur.wait_stream(self.s_finish)
        self._last_finish.clear()
        self._queued.clear()
        self._prev_finish = None
