"""Debug aid: decode one file through the device path, report where the planes differ from the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from webp_amd import parse
from webp_amd.decoder import LossyBatch
from webp_amd.device import DeviceLossyBatch
path = sys.argv[1]
data = open(path, "rb").read()
fr = parse(data).frames[0]
db = DeviceLossyBatch(LossyBatch([fr]), "cuda:0")
db.recon_filter(); db.finish(); torch.cuda.synchronize()
want = oracle.decode(data)
y, u, v = db.planes(0)
for name, g, w, stride in (("y", y, want["y"], 16 * fr.mbw), ("u", u, want["u"], 8 * fr.mbw), ("v", v, want["v"], 8 * fr.mbw)):
    bad = np.flatnonzero(g != w)
    print(name, "diffs:", bad.size)
    if bad.size:
        rows, cols = bad // stride, bad % stride
        mb = 16 if name == "y" else 8
        print("  rows%%mb:", np.unique(rows % mb)[:20], "cols%%mb:", np.unique(cols % mb)[:20])
        print("  mb rows:", np.unique(rows // mb)[:30], "mb cols:", np.unique(cols // mb)[:40])
        for b in bad[:8]:
            print("   at row %d col %d got %d want %d" % (b // stride, b % stride, g[b], w[b]))
