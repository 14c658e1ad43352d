import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
from webp_amd._lib import lib
files, batch, db, paths, _ = bench.build_batch(1024, "lossy1080_*.webp", "cuda:0")
db.recon_filter(); torch.cuda.synchronize()
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for ov in (1, 0):
    lib().wp_hint_finish_overlap(ov)
    print(os.environ.get("WEBP_AMD_LIB","in-tree").split("/")[-1], "overlap-hint", ov, "finish alone %.3f ms" % t(db.finish), flush=True)
