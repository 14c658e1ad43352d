import sys, ctypes, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from webp_amd._lib import AlphaFrame, lib, check
from oracle import oracle
n, w, h = 300, 1280, 720
rng = np.random.default_rng(1)
base = (rng.integers(0, 256, w * h, dtype=np.uint8) // 8).astype(np.uint8)
lib().wp_init(0)
for f in (3, 1, 2):
    desc = (AlphaFrame * n)()
    off = 0
    for i in range(n):
        d = desc[i]; d.width, d.height, d.filter, d.src_is_vp8l_rgba, d.src_off, d.dst_off = w, h, f, 0, off, off + (0 if f != 1 else 0)
        off += w * h
    src = torch.from_numpy(np.tile(base, n)).cuda(); dst = torch.empty_like(src)
    dd = torch.from_numpy(np.frombuffer(bytes(desc), dtype=np.uint8).copy()).cuda()
    def run(): check(lib().wp_alpha_unfilter_dev(n, desc, dd.data_ptr(), src.data_ptr(), dst.data_ptr(), None))
    run(); torch.cuda.synchronize()
    assert np.array_equal(dst[:w * h].cpu().numpy(), oracle.unfilter_alpha(base, w, f)), f
    assert np.array_equal(dst[-w * h:].cpu().numpy(), oracle.unfilter_alpha(base, w, f)), f
    t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
    print("filter %d: %d x %dx%d unfilterAlpha %.3f ms, bit-exact" % (f, n, w, h, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
