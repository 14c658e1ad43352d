"""Experiment: successive batches software-pipelined over HIP streams.  Variant 2: finishing pass of step k on stream B while
stream A reconstructs step k+1 (one workspace).  Variant 3: recon / filter / finish each on a stream of its own, two
device batches alternating (double-buffered workspace and outputs).   python tools/overlap_times.py [images] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from webp_amd.device import DeviceLossyBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
files, batch, db, paths, _ = bench.build_batch(n, "lossy1080_*.webp", "cuda:0")
db2 = DeviceLossyBatch(batch, "cuda:0")
dbs = [db, db2]
px = sum(batch.desc[i].width * batch.desc[i].height for i in range(batch.n))
cur = torch.cuda.current_stream()
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    for _ in range(steps):
        db.decode_to_argb()


def piped2():
    sA.wait_stream(cur); sB.wait_stream(cur)
    fin = None
    for _ in range(steps):
        with torch.cuda.stream(sA):
            db.recon(planes=False)
            if fin is not None:
                sA.wait_event(fin)              # the filter rewrites the S tiles the previous finishing pass reads
            db.loopfilter(planes=False)
            filt = torch.cuda.Event(); filt.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(filt)
            db.finish()
            fin = torch.cuda.Event(); fin.record(sB)
    cur.wait_stream(sA); cur.wait_stream(sB)


def piped3(two_streams=False):
    for s in (sA, sB, sC):
        s.wait_stream(cur)
    s_filter = sA if two_streams else sB
    fin = [None, None]                          # finishing pass of the last step that used workspace i
    for k in range(steps):
        d = dbs[k & 1]
        with torch.cuda.stream(sA):
            if fin[k & 1] is not None:
                sA.wait_event(fin[k & 1])       # workspace and outputs of step k-2 are free again
            d.recon(planes=False)
            rec = torch.cuda.Event(); rec.record(sA)
        with torch.cuda.stream(s_filter):
            s_filter.wait_event(rec)
            d.loopfilter(planes=False)
            filt = torch.cuda.Event(); filt.record(s_filter)
        with torch.cuda.stream(sC):
            sC.wait_event(filt)
            d.finish()
            fin[k & 1] = torch.cuda.Event(); fin[k & 1].record(sC)
    for s in (sA, sB, sC):
        cur.wait_stream(s)


from webp_amd.device import LossyPipeline


def helper(prio_main, prio_fin):
    pipe = LossyPipeline("cuda:0")
    pipe.s_main = torch.cuda.Stream(priority=prio_main)
    pipe.s_finish = torch.cuda.Stream(priority=prio_fin)
    pipe.s_main.wait_stream(cur); pipe.s_finish.wait_stream(cur)
    def f():
        for _ in range(steps):
            pipe.submit(db)
        pipe.drain()
    return f


for name, fn in (("helper prio 0/0", helper(0, 0)), ("helper finish high", helper(0, -1)), ("helper main high", helper(-1, 0)),
                 ("helper prio 0/0", helper(0, 0)), ("helper finish high", helper(0, -1)), ("helper main high", helper(-1, 0))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-26s %.3f ms per step, %.1f Gpix/s" % (name, 1e3 * dt / steps, px * steps / dt / 1e9), flush=True)

for name, fn in (("serial", serial), ("2 streams, 1 workspace", piped2), ("3 streams, 2 workspaces", piped3),
                 ("2 streams, 2 workspaces", lambda: piped3(True)), ("serial", serial), ("2 streams, 1 workspace", piped2),
                 ("3 streams, 2 workspaces", piped3), ("2 streams, 2 workspaces", lambda: piped3(True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-26s %.3f ms per step, %.1f Gpix/s" % (name, 1e3 * dt / steps, px * steps / dt / 1e9), flush=True)
