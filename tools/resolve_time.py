"""Times wp_vp8l_resolve_dev (decodePix's pixel loop on the device) on batches of real token streams and prints, beside
it, what the same loop costs inside the host parse (parse with and without WP_PARSE_VP8L_TOKENS, one thread)."""
import ctypes, glob, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from webp_amd._lib import Vp8lTokensFrame, check, lib
from webp_amd.decoder import parse

L = lib()
check(L.wp_init(0))
files = [os.path.join(ROOT, "tests/golden/inputs/3ll.webp"), os.path.join(ROOT, "tests/golden/inputs/5ll.webp")] + \
        sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossless1080_*.webp")))
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) > 2:                 # force the kernel's LDS ring: python tools/resolve_time.py 2048 4096
    check(L.wp_debug_set_resolve_ring(int(sys.argv[2])))
    print("ring forced to", sys.argv[2])
for path in files:
    data = open(path, "rb").read()
    t0 = time.perf_counter(); parse(data); t1 = time.perf_counter(); pf = parse(data, vp8l_tokens=True); t2 = time.perf_counter()
    f = pf.frames[0]
    tk = f.buf(6, np.uint32)
    w, h = f.coded_width, f.vp8l_height
    desc = (Vp8lTokensFrame * copies)()
    stride = (4 * w * h + 255) // 256 * 256
    for i in range(copies):
        d = desc[i]
        d.coded_width, d.height, d.cc_bits, d.tok_off, d.n_words, d.pix_off = w, h, f.cc_bits, 0, tk.size, stride * i   # every image reads the one stream
    tok = torch.from_numpy(tk.view(np.int32).copy()).cuda()
    dd = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).cuda()
    pix = torch.empty(stride * copies, dtype=torch.uint8, device="cuda")
    run = lambda: check(L.wp_vp8l_resolve_dev(copies, desc, dd.data_ptr(), tok.data_ptr(), pix.data_ptr(), None))
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-28s %4dx%-4d tokens %8d words  host parse %.1f ms, with tokens %.1f ms | device loop, %d images: %.2f ms = %.1f Gpix/s (%.3f ms per image if alone on a wave)"
          % (os.path.basename(path), w, h, tk.size, (t1 - t0) * 1e3, (t2 - t1) * 1e3, copies, ms, copies * w * h / ms / 1e6, ms / max(1, -(-copies // 512))))
