"""Device-resident timing of the VP8L inverse-transform chain (cfg 3: 1080p lossless) and its per-kernel split:
  python tools/lossless_times.py [images]      (run under rocprofv3 --kernel-trace --stats for the per-kernel table)
  python tools/lossless_times.py [images] [stream index]   one of the bench streams only"""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webp_amd import parse
from webp_amd._lib import Vp8lFrame, check, lib
from webp_amd.decoder import LosslessBatch

pattern = "lossless1080_*.webp"
if "--pattern" in sys.argv:           # other streams of bench_data/: python tools/lossless_times.py 1024 --pattern "ll1080_mixed_*.webp"
    k = sys.argv.index("--pattern")
    pattern = sys.argv[k + 1]
    del sys.argv[k:k + 2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(root, "bench_data", pattern)))
if len(sys.argv) > 2:                # one of the streams only: python tools/lossless_times.py 1024 <index>
    paths = [paths[int(sys.argv[2])]]
files = [parse(open(p, "rb").read()) for p in paths]
frames = [files[i % len(files)].frames[0] for i in range(n)]
b = LosslessBatch(frames)
dev = torch.device("cuda:0")
lib().wp_init(0)
d_frames = torch.from_numpy(np.frombuffer(bytes(b.desc), dtype=np.uint8).copy()).to(dev)
pix0 = torch.from_numpy(b.pix).to(dev)
pix = torch.empty_like(pix0)
aux = torch.from_numpy(b.aux).to(dev)
tmp = torch.empty(b.out_bytes, dtype=torch.uint8, device=dev)
out = torch.empty(b.out_bytes, dtype=torch.uint8, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
px = sum(b.desc[i].width * b.desc[i].height for i in range(n))
ts = []
for it in range(6):
    pix.copy_(pix0)                 # the chain consumes its input (transforms run in place, as in the reference)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().wp_vp8l_inverse_dev(n, b.desc, d_frames.data_ptr(), pix.data_ptr(), aux.data_ptr(), tmp.data_ptr(), out.data_ptr(), 1, stream))
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
t = min(ts[1:])
kinds = {}
for f in files:
    for (ty, bits, ow) in f.frames[0].transforms:
        kinds[(ty, bits)] = kinds.get((ty, bits), 0) + 1
print("%d x 1080p lossless (%d streams, transforms (type,bits): %s): %.3f ms per batch, %.1f Gpix/s, %.0f GB/s at 8 B/px algorithmic"
      % (n, len(files), sorted(kinds), t, px / t / 1e6, 8 * px / t / 1e6))
