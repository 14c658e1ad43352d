"""Experiment: in the two-stream step pipeline, start the reconstruction of step k+1 (or the finishing pass of step k) a
little later than the other, to see how the split of the GPU between the two depends on who is dispatched first.
python tools/overlap_delay.py [images]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from webp_amd.device import LossyPipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = 10
files, batch, db, paths, _ = bench.build_batch(n, "lossy1080_*.webp", "cuda:0")
px = sum(batch.desc[i].width * batch.desc[i].height for i in range(batch.n))
CYC = 2.1e6          # torch.cuda._sleep counts clock cycles; ~2.1 GHz under load


class Delayed(LossyPipeline):
    def __init__(self, d_recon, d_finish):
        super().__init__("cuda:0")
        self.d_recon, self.d_finish = d_recon, d_finish
        self._last_finish = {}          # this experiment keeps the single-workspace schedule LossyPipeline had before it alternated workspaces

    def submit(self, db):
        fin = self._last_finish.get(id(db))
        with torch.cuda.stream(self.s_main):
            if self.d_recon:
                torch.cuda._sleep(int(self.d_recon * CYC))
            db.recon(planes=False)
            if fin is not None:
                self.s_main.wait_event(fin)
            db.loopfilter(planes=False)
            filtered = torch.cuda.Event(); filtered.record(self.s_main)
        with torch.cuda.stream(self.s_finish):
            self.s_finish.wait_event(filtered)
            if self.d_finish:
                torch.cuda._sleep(int(self.d_finish * CYC))
            db.finish()
            done = torch.cuda.Event(); done.record(self.s_finish)
        self._last_finish[id(db)] = done
        return done


for dr, df in ((0, 0), (0.3, 0), (0.6, 0), (1.0, 0), (1.5, 0), (0, 0.3), (0, 0.6), (0, 1.0), (0, 0)):
    p = Delayed(dr, df)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            p.submit(db)
        p.drain(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("delay recon %.1f ms, finish %.1f ms: %.3f ms per step, %.1f Gpix/s" % (dr, df, 1e3 * best / steps, px * steps / best / 1e9), flush=True)
