"""Times the lossy stages (and variants of the finishing pass) with HIP events: python tools/stage_times.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
files, batch, db, paths, _ = bench.build_batch(n, "lossy1080_*.webp", "cuda:0")


def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters


db.decode_to_argb()
print("recon            %.3f ms" % t(lambda: db.recon(planes=False)))
print("filter           %.3f ms" % t(lambda: db.loopfilter(planes=False)))
print("finish all       %.3f ms" % t(lambda: db.finish(True, True)))
print("finish argb only %.3f ms" % t(lambda: db.finish(False, True)))
print("finish planes    %.3f ms" % t(lambda: db.finish(True, False)))
print("convert (linear) %.3f ms" % t(lambda: db.convert()))
print("fused            %.3f ms" % t(lambda: db.decode_to_argb()))
db.argb.zero_(); db.decode_to_argb(); torch.cuda.synchronize()
print("checksum argb %d  y %d" % (int(db.argb[::4099].to(torch.int64).sum()), int(db.y[::4099].to(torch.int64).sum())))
