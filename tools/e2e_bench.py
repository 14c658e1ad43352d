"""End-to-end throughput (file bytes in host memory -> ARGB + planes in pinned host memory) for batched 1080p lossy WebP.
Reported separately from bench.py, whose `value` is the device-resident rate.
  staged   : threaded host parse (wp_parse_batch) of everything, batch assembly, H2D, device pipeline, D2H, one after the other
  streamed : webp_amd.streaming.LossyStream -- the same stages overlapped over chunks of files
python tools/e2e_bench.py [images] [threads] [chunk]"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webp_amd import parse_many
from webp_amd.decoder import LossyBatch
from webp_amd.device import DeviceLossyBatch
from webp_amd.streaming import LossyStream

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 32
paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_data", "lossy1080_*.webp")))
datas = [open(paths[i % len(paths)], "rb").read() for i in range(n)]
torch.cuda.init(); torch.zeros(1, device="cuda:0")
pinned = {}


def to_host(name, t):
    """D2H into a pinned staging buffer that is allocated once (what a serving loop would keep)."""
    if name not in pinned or pinned[name].numel() < t.numel():
        pinned[name] = torch.empty(t.numel(), dtype=torch.uint8, pin_memory=True)
    pinned[name][:t.numel()].copy_(t, non_blocking=True)


for rep in range(2):
    t0 = time.perf_counter()
    files = parse_many(datas, threads)
    t1 = time.perf_counter()
    batch = LossyBatch([f.frames[0] for f in files], threads=threads)
    t2 = time.perf_counter()
    db = DeviceLossyBatch(batch, "cuda:0")
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    db.decode_to_argb()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    to_host('argb', db.argb); to_host('y', db.y); to_host('u', db.u); to_host('v', db.v)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    px = sum(batch.desc[i].width * batch.desc[i].height for i in range(batch.n))
    print("staged rep %d: %d images, %d threads: parse %.3f s (%.2f ms/image/thread), assemble %.3f, alloc+H2D %.3f, device %.4f, D2H %.3f  => %.1f Mpix/s end to end (parse only: %.1f Mpix/s)"
          % (rep, n, threads, t1 - t0, 1e3 * (t1 - t0) * threads / n, t2 - t1, t3 - t2, t4 - t3, t5 - t4, px / (t5 - t0) / 1e6, px / (t1 - t0) / 1e6), flush=True)
    del db, files, batch
pinned.clear()
torch.cuda.empty_cache()

stream = LossyStream("cuda:0", threads=threads, chunk=chunk, planes=True)
for rep in range(3):
    t0 = time.perf_counter()
    out = stream.decode(datas)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streamed rep %d: %d images, %d threads, chunks of %d: %.3f s => %.1f Mpix/s end to end" % (rep, n, threads, chunk, dt, px / dt / 1e6), flush=True)
    chk = int(out[0][::40009].to(torch.int64).sum())
    del out
print("checksum", chk)
