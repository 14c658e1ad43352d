"""cfg 4 (BASELINE.json configs[3]): 300 x 1280x720 lossy frames with a VP8L-coded ALPH plane, device resident:
alpha chain (last pass writes the green channel straight into the alpha slab) + unfilterAlpha in place + reconstruction + loop
filter + finishing pass with alpha.  Checks the first frames against the oracle, then times the step.
   python tools/alpha_times.py [frames]"""
import ctypes, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from webp_amd import parse
from webp_amd._lib import AlphaFrame, check, lib
from webp_amd.decoder import LossyBatch, LosslessBatch, _load_vp8l_meta
from webp_amd.device import DeviceLossyBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sorted(glob.glob(os.path.join(root, "bench_data", "lossy720_alpha_*.webp")))
datas = [open(p, "rb").read() for p in paths]
files = [parse(d) for d in datas]
frames = [files[i % len(files)].frames[0] for i in range(n)]
for f in frames:
    _load_vp8l_meta(f)
    assert f.has_alpha and f.alpha_is_vp8l, "this tool expects VP8L-coded ALPH planes"
batch = LossyBatch(frames, with_alpha=[True] * n)
lb = LosslessBatch(frames, alpha_image=True)
for i in range(n):
    lb.desc[i].out_off = batch.desc[i].alpha_off          # the chain's green-only last pass writes the alpha slab itself
dev = torch.device("cuda:0")
db = DeviceLossyBatch(batch, dev, alpha=np.zeros(max(batch.alpha_bytes, 16), np.uint8))
d_lf = torch.from_numpy(np.frombuffer(bytes(lb.desc), dtype=np.uint8).copy()).to(dev)
pix0 = torch.from_numpy(lb.pix).to(dev); pix = torch.empty_like(pix0)
aux = torch.from_numpy(lb.aux).to(dev)
tmp = torch.empty(lb.out_bytes, dtype=torch.uint8, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
adesc = (AlphaFrame * n)()                                 # unfilterAlpha in place on the alpha slab (skipped when every ALPH filter is 0)
for i, f in enumerate(frames):
    a = adesc[i]
    a.width, a.height, a.filter, a.src_is_vp8l_rgba = f.alpha_width, f.alpha_height, f.alpha_filter, 0
    a.src_off = a.dst_off = batch.desc[i].alpha_off
d_af = torch.from_numpy(np.frombuffer(bytes(adesc), dtype=np.uint8).copy()).to(dev)
need_unfilter = any(f.alpha_filter != 0 for f in frames)

def step():
    pix.copy_(pix0)                                        # the chain consumes its input
    check(lib().wp_vp8l_inverse_dev(n, lb.desc, d_lf.data_ptr(), pix.data_ptr(), aux.data_ptr(), tmp.data_ptr(), db.alpha.data_ptr(), 2, stream))
    if need_unfilter:
        check(lib().wp_alpha_unfilter_dev(n, adesc, d_af.data_ptr(), db.alpha.data_ptr(), db.alpha.data_ptr(), stream))
    db.decode_to_argb()

step(); torch.cuda.synchronize()
from oracle import oracle
for i in range(min(n, len(files))):
    want = oracle.decode(datas[i])
    assert np.array_equal(db.argb_of(i), want["to_argb"]), "ARGB (with alpha) of frame %d differs from the oracle" % i
print("first %d frames bit-exact against the oracle (A,R,G,B with the decoded alpha)" % min(n, len(files)), flush=True)
ts = []
for it in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); step(); e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1))
t = min(ts[1:])
px = sum(batch.desc[i].width * batch.desc[i].height for i in range(n))
print("%d x %dx%d lossy + VP8L alpha: %.3f ms per batch (incl. the %.1f MB input copy the chain needs), %.1f Gpix/s"
      % (n, batch.desc[0].width, batch.desc[0].height, t, pix0.numel() / 1e6, px / t / 1e6))
