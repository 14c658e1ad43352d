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

    def second_workspace(self):
        """A second workspace of the same size (LossyPipeline alternates between the two)."""
        return self.workspace(1)

    def workspace(self, k):
        """Workspace number k (0 = the batch's own); further ones are allocated on first use (LossyPipeline rotates them)."""
        if k == 0:
            return self.work
        extra = self.__dict__.setdefault("_extra_work", {})
        if k not in extra:
            extra[k] = torch.empty(max(self.work_bytes, 256), dtype=torch.uint8, device=self.device)
        return extra[k]

    def recon(self, planes=True, work=None):
        """Reconstruction into the tiled workspace; planes=True also writes the unfiltered row-major planes."""
        work = self.work if work is None else work
        check(lib().wp_vp8_recon_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.mb.data_ptr(), self.coeff.data_ptr(),
                                     work.data_ptr(), self.work_bytes, *self._planes(planes), self._stream()))

    def loopfilter(self, planes=True, work=None):
        """planes=True: filter the row-major planes in place; planes=False: workspace R tiles -> S tiles."""
        work = self.work if work is None else work
        check(lib().wp_vp8_loopfilter_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.filt.data_ptr(),
                                          work.data_ptr(), self.work_bytes, *self._planes(planes), self._stream()))

    def recon_filter(self, work=None):
        """Reconstruction and loop filter (workspace R tiles -> S tiles) as one launch."""
        work = self.work if work is None else work
        check(lib().wp_vp8_recon_filter_dev(self.b.n, self.b.desc, self.frames.data_ptr(), self.mb.data_ptr(), self.filt.data_ptr(),
                                            self.coeff.data_ptr(), work.data_ptr(), self.work_bytes, self._stream()))

    def finish(self, planes=True, argb=True, work=None, first=0, count=None):
        """Finishing pass of frames [first, first + count) (default: all)."""
        work = self.work if work is None else work
        count = self.b.n - first if count is None else count
        check(lib().wp_vp8_finish_range_dev(self.b.n, self.b.desc, self.frames.data_ptr(), first, count, work.data_ptr(), self.work_bytes,
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


_PIPELINE_STREAMS = {}


def pipeline_streams(device):
    """The two HIP streams every LossyPipeline of this process uses on `device` (default priority), created once.  A
    process therefore never holds more than two extra streams per device however many pipelines it builds, and none with a
    priority: round 1's per-object streams, one of them high-priority, once stalled for good in a process that had created
    many ("the round-1 hang")."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _PIPELINE_STREAMS:
        _PIPELINE_STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _PIPELINE_STREAMS[key]


class LossyPipeline:
    """Software pipeline over successive device batches, in two phases per step on two HIP streams:

        phase 1:  reconstruction of step k     |  finishing pass of step k-1, frames [0, split)
        phase 2:  loop filter of step k        |  finishing pass of step k-1, frames [split, n)

    The finishing pass is bound by memory bandwidth, reconstruction and the loop filter by instruction issue: side by side
    they fill each other's gaps.  The finishing pass is split because next to reconstruction alone the pair is already at
    the bandwidth the device delivers (23 GB in 5.0 ms), while the loop filter that followed ran with the memory system
    half idle.  The two kernels of a phase are released together, after both kernels of the phase before have finished:
    a persistent kernel (reconstruction, loop filter) that starts while a streaming kernel holds the GPU gets its
    workgroups placed piecemeal and runs far slower (tools/overlap_delay.py).

    Steps alternate between two workspaces (`DeviceLossyBatch.second_workspace`), so the kernels of step k never touch the
    tiles the finishing pass of step k-1 reads.  The finishing pass of a step is enqueued by the NEXT `submit()` (or by
    `drain()`): `submit()` returns the event of the previous step's outputs (None for the first), `drain()` completes
    everything.

    one_launch=True (what bench.py uses): reconstruction + loop filter are ONE kernel (wp_vp8_recon_filter_dev) and the whole
    finishing pass of step k-1 runs beside it.

    Both streams have the default priority and are shared by all pipelines of the process (`pipeline_streams`), on
    purpose: see DESIGN.md, "the recorded hang".  At most `max_in_flight` steps are queued ahead of the GPU."""

    def __init__(self, device="cuda:0", max_in_flight=3, split=0.6, finish_delay_us=40, one_launch=False, workspaces=2):
        self.device = torch.device(device)
        self.s_main, self.s_finish = pipeline_streams(self.device)
        self.max_in_flight = max_in_flight
        self.split = split
        self.workspaces = max(2, workspaces)
        self.one_launch = one_launch    # reconstruction + loop filter as one launch (wp_vp8_recon_filter_dev), the whole finishing pass of the step before beside it
        self.finish_delay_cycles = int(finish_delay_us * 2100)     # torch.cuda._sleep counts clock cycles (~2.1 GHz under load)
        self._queued = []               # events of the steps not yet known to be complete, oldest first
        self._pending = None            # (db, work, planes, argb, filtered event, timing list) of the step whose finishing pass is not enqueued yet
        self._phase2_done = None        # event: the most recent second-half finishing pass
        self._k = 0
        cur = torch.cuda.current_stream(self.device)
        self.s_main.wait_stream(cur)
        self.s_finish.wait_stream(cur)

    @staticmethod
    def _mark(stream, timing):
        if timing is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream)
        return e

    def _let_persistent_kernel_start_first(self):
        # Both kernels of a phase become runnable at the same moment; the one whose stream saw the phase before end last is
        # dispatched first.  If that is the streaming finishing pass, the persistent kernel beside it finds the GPU
        # occupied and runs up to twice as long, so the finishing pass is held back by a few tens of microseconds.
        if self.finish_delay_cycles > 0:
            torch.cuda._sleep(self.finish_delay_cycles)

    def _event(self, stream):
        e = torch.cuda.Event()
        e.record(stream)
        return e

    def submit(self, db: "DeviceLossyBatch", planes=True, argb=True, timing=None):
        """Enqueues reconstruction and loop filter of one step and, beside them, the finishing pass of the step before.
        `timing`: a list that receives (recon start, recon end, filter start, filter end, finish start, finish end)
        timing events of THIS step once its finishing pass has been enqueued (by the next submit() or drain())."""
        while len(self._queued) >= self.max_in_flight:
            self._queued.pop(0).synchronize()
        work = db.workspace(self._k % self.workspaces)
        self._k += 1
        prev = self._pending
        marks = {}
        if self.one_launch:
            with torch.cuda.stream(self.s_main):
                # the finishing pass that last read the workspace this launch is about to overwrite
                hist = self.__dict__.setdefault("_finish_hist", [])
                if len(hist) >= self.workspaces - 1:
                    self.s_main.wait_event(hist[-(self.workspaces - 1)])
                marks["r0"] = self._mark(self.s_main, timing)
                db.recon_filter(work=work)
                marks["r1"] = self._mark(self.s_main, timing)
                marks["f0"] = marks["f1"] = marks["r1"]
                filtered = self._event(self.s_main)
            done_prev = None
            if prev is not None:
                pdb, pwork, pplanes, pargb, pfiltered, ptiming, pmarks = prev
                with torch.cuda.stream(self.s_finish):
                    self.s_finish.wait_event(pfiltered)
                    self._let_persistent_kernel_start_first()
                    pmarks["c0"] = self._mark(self.s_finish, ptiming)
                    pdb.finish(pplanes, pargb, work=pwork)
                    pmarks["c1"] = self._mark(self.s_finish, ptiming)
                    done_prev = self._event(self.s_finish)
                self._phase2_done = done_prev
                self._finish_hist.append(done_prev)
                del self._finish_hist[:-max(4, self.workspaces)]      # the wait above looks workspaces - 1 entries back
                self._queued.append(done_prev)
                if ptiming is not None:
                    ptiming.append(pmarks)
            self._pending = (db, work, planes, argb, filtered, timing, marks)
            return done_prev
        # ---- phase 1 ----
        with torch.cuda.stream(self.s_main):
            if self._phase2_done is not None:
                self.s_main.wait_event(self._phase2_done)       # the GPU is empty when the pair starts
            marks["r0"] = self._mark(self.s_main, timing)
            db.recon(planes=False, work=work)
            marks["r1"] = self._mark(self.s_main, timing)
            recon_done = self._event(self.s_main)
        half_a = None
        if prev is not None:
            pdb, pwork, pplanes, pargb, pfiltered, ptiming, pmarks = prev
            n_a = min(pdb.b.n, max(0, int(round(pdb.b.n * self.split))))
            with torch.cuda.stream(self.s_finish):
                self.s_finish.wait_event(pfiltered)
                self._let_persistent_kernel_start_first()
                pmarks["c0"] = self._mark(self.s_finish, ptiming)
                pdb.finish(pplanes, pargb, work=pwork, first=0, count=n_a)
                pmarks["ca"] = self._mark(self.s_finish, ptiming)
                half_a = self._event(self.s_finish)
        # ---- phase 2 ----
        with torch.cuda.stream(self.s_main):
            if half_a is not None:
                self.s_main.wait_event(half_a)
            marks["f0"] = self._mark(self.s_main, timing)
            db.loopfilter(planes=False, work=work)
            marks["f1"] = self._mark(self.s_main, timing)
            filtered = self._event(self.s_main)
        done_prev = None
        if prev is not None:
            with torch.cuda.stream(self.s_finish):
                self.s_finish.wait_event(recon_done)
                self._let_persistent_kernel_start_first()
                pmarks["cb"] = self._mark(self.s_finish, ptiming)
                pdb.finish(pplanes, pargb, work=pwork, first=n_a, count=pdb.b.n - n_a)
                pmarks["c1"] = self._mark(self.s_finish, ptiming)
                done_prev = self._event(self.s_finish)
            self._phase2_done = done_prev
            self._queued.append(done_prev)
            if ptiming is not None:
                ptiming.append(pmarks)
        self._pending = (db, work, planes, argb, filtered, timing, marks)
        return done_prev

    def flush(self):
        """Enqueues the finishing pass of the last submitted step (nothing left to overlap it with)."""
        if self._pending is None:
            return None
        pdb, pwork, pplanes, pargb, pfiltered, ptiming, pmarks = self._pending
        self._pending = None
        with torch.cuda.stream(self.s_finish):
            self.s_finish.wait_event(pfiltered)
            pmarks["c0"] = self._mark(self.s_finish, ptiming)
            pdb.finish(pplanes, pargb, work=pwork)
            pmarks["c1"] = self._mark(self.s_finish, ptiming)
            done = self._event(self.s_finish)
        if ptiming is not None:
            ptiming.append(pmarks)
        self._phase2_done = done
        self._queued.append(done)
        return done

    def drain(self):
        self.flush()
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.s_main)
        cur.wait_stream(self.s_finish)
        self._queued.clear()
        self._phase2_done = None


class NativePipeline:
    """The C ABI's software pipeline (wp_vp8_pipeline_*, csrc/pipeline.cpp): what LossyPipeline(one_launch=True) does, but
    inside the library -- its own two streams, its own workspaces, its own hold-off -- so that a caller in any language gets
    the pipelined step.  bench.py's timed region runs through this object.

    submit(db) enqueues the reconstruction + loop filter kernel of `db` and, beside it, the finishing pass of the batch
    submitted before; drain() completes everything (host sync); flush() makes the current torch stream wait instead."""

    def __init__(self, device="cuda:0", work_bytes=0, workspaces=2, timing=False, hold_us=None):
        self.device = torch.device(device)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().wp_vp8_pipeline_create(workspaces, work_bytes, 1 if timing else 0, ctypes.byref(self._h)))
        if hold_us is not None:
            check(lib().wp_vp8_pipeline_set_hold_us(self._h, int(hold_us)))
        self._keep = []

    def submit(self, db: "DeviceLossyBatch", planes=True, argb=True, after_current_stream=True, alpha_ready=None):
        """after_current_stream: the kernel waits for what has been enqueued on the current torch stream so far (uploads).
        alpha_ready: a torch.cuda.Event only the finishing pass waits for (the alpha planes, produced on another stream)."""
        ev = None
        if after_current_stream:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        y, u, v = db._planes(planes)
        check(lib().wp_vp8_pipeline_submit(self._h, db.b.n, db.b.desc, db.frames.data_ptr(), db.mb.data_ptr(), db.filt.data_ptr(),
                                           db.coeff.data_ptr(), y, u, v, db._alpha(), db.argb.data_ptr() if argb else None,
                                           ctypes.c_void_p(ev.cuda_event) if ev is not None else None,
                                           ctypes.c_void_p(alpha_ready.cuda_event) if alpha_ready is not None else None))
        self._keep = (self._keep + [(db, ev, alpha_ready)])[-8:]          # buffers and the event stay alive while steps are in flight

    def flush(self):
        check(lib().wp_vp8_pipeline_flush(self._h, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def drain(self):
        check(lib().wp_vp8_pipeline_drain(self._h))
        self._keep = []

    def stats(self, reset=False):
        """(ms in reconstruction + loop filter launches, ms in finishing launches, completed steps) since the last reset."""
        a, b, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_long(0)
        check(lib().wp_vp8_pipeline_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n), 1 if reset else 0))
        return a.value, b.value, n.value

    def close(self):
        if self._h:
            lib().wp_vp8_pipeline_destroy(self._h)
            self._h = ctypes.c_void_p()
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
