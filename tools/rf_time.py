"""One-kernel reconstruction + loop filter (wp_vp8_recon_filter_dev) against the two launches: parity of the finished
outputs and time.  python tools/rf_time.py [images] [pattern]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pattern = sys.argv[2] if len(sys.argv) > 2 else "lossy1080_*.webp"
files, batch, db, paths, _ = bench.build_batch(n, pattern, "cuda:0")
db.recon(planes=False); db.loopfilter(planes=False); db.finish(); torch.cuda.synchronize()
ref = [t.clone() for t in (db.argb, db.y, db.u, db.v)]
for t in (db.argb, db.y, db.u, db.v): t.zero_()
db.recon_filter(); db.finish(); torch.cuda.synchronize()
ok = all(torch.equal(a, b) for a, b in zip((db.argb, db.y, db.u, db.v), ref))
print("one-kernel recon+filter equals staged:", ok, flush=True)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
print("recon %.3f  filter %.3f  recon_filter %.3f  finish %.3f ms" % (t(lambda: db.recon(planes=False)), t(lambda: db.loopfilter(planes=False)), t(db.recon_filter), t(db.finish)), flush=True)
