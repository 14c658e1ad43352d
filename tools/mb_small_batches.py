import sys, os, glob
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, torch
from make_anim import build_anim
from webp_amd import parse
from webp_amd.decoder import LossyBatch
from webp_amd.device import DeviceLossyBatch
stills = [open(p, "rb").read() for p in sorted(glob.glob("/root/repo/bench_data/lossy720_alpha_*.webp"))]
pf = parse(build_anim(stills, 512))
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for n in (1, 8, 44, 128, 200, 256, 300, 384, 512):
    db = DeviceLossyBatch(LossyBatch(pf.frames[:n]), "cuda:0")
    print(n, "frames: recon_filter %.3f ms  finish %.3f ms" % (t(db.recon_filter), t(db.finish)), flush=True)
    del db
