"""Recon / filter / finish time per stream of bench_data (1024 copies of one stream each): shows how much the cost of a
frame depends on its content.   python tools/stream_times.py [images]"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def t(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / iters


pats = ["lossy1080_*.webp", "lossy1080_q75_[0-7].webp", "lossy1080_q75_0.webp", "lossy1080_q90_0.webp", "lossy1080_q50_sharp3_2.webp", "lossy1080_q75_simple_1.webp"]
for pat in pats:
    files, batch, db, paths, _ = bench.build_batch(n, pat, "cuda:0")
    db.decode_to_argb()
    print("%-32s recon %.3f  filter %.3f  finish %.3f ms   (%.1f coeff blocks/MB)" % (pat, t(lambda: db.recon(planes=False)), t(lambda: db.loopfilter(planes=False)),
          t(lambda: db.finish()), batch.n_coeff_blocks / batch.n_mb), flush=True)
    del db, batch, files
    torch.cuda.empty_cache()
