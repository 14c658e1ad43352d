"""Where the host time of LossyStream.decode goes (monkey-patched timers).  python tools/e2e_profile.py [images] [threads] [chunk]"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import webp_amd.streaming as st
from webp_amd import decoder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 128
paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "lossy1080_*.webp")))
datas = [open(paths[i % len(paths)], "rb").read() for i in range(n)]
torch.zeros(1, device="cuda:0")
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w


decoder.ParseJob.take = timed("take (wait for parse + wrap)", decoder.ParseJob.take)
st.LossyBatch.__init__ = timed("LossyBatch (layout + gather)", st.LossyBatch.__init__)
st._Slot.pinned = timed("pinned alloc", st._Slot.pinned)
st._Slot.on_device = timed("device alloc", st._Slot.on_device)
s = st.LossyStream("cuda:0", threads=threads, chunk=chunk)
for rep in range(3):
    acc.clear()
    t0 = time.perf_counter(); out = s.decode(datas); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rep %d: %.3f s total; " % (rep, dt) + "; ".join("%s %.3f" % kv for kv in acc.items()), flush=True)
    del out
