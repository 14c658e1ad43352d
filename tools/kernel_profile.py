"""Per-phase cycle profile of one wavefront (workgroup 0, wave 0) of the recon kernel.
Needs a -DWP_PROFILE build: WEBP_AMD_LIB=webp_amd/variants/libwebp_amd_prof.so python tools/kernel_profile.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from webp_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
files, batch, db, paths, _ = bench.build_batch(n, "lossy1080_*.webp", "cuda:0")
L = lib()
buf = (ctypes.c_ulonglong * 32)()
db.recon(planes=False); torch.cuda.synchronize()
L.wp_debug_profile(buf, 1)
db.recon(planes=False); torch.cuda.synchronize()
L.wp_debug_profile(buf, 0)
def show(names):
    tot = sum(buf[i] for i in range(10))
    for i, nm in enumerate(names):
        print("%-18s %12d  %5.1f%%" % (nm, buf[i], 100.0 * buf[i] / max(tot, 1)))
    print("total cycles", tot)


names = ["0 wait+top loads", "1 residuals", "2 tile prep", "3 i16", "4 i4 loop", "5 chroma", "6 publish+store", "7 band end", "8 loop top", "9 band start"]
print("== recon"); show(names)
db.loopfilter(planes=False); torch.cuda.synchronize()
L.wp_debug_profile_filter(buf, 1)
db.loopfilter(planes=False); torch.cuda.synchronize()
L.wp_debug_profile_filter(buf, 0)
print("== filter")
show(["0 wait+prefetch", "1 tile fill", "2 V edges", "3 H edges", "4 publish", "5 emit+handdown", "6", "7", "8 loop top", "9 band start"])

db.recon_filter(); torch.cuda.synchronize()
L.wp_debug_profile_mb(buf, 1)
db.recon_filter(); torch.cuda.synchronize()
L.wp_debug_profile_mb(buf, 0)
print("== one-kernel path (vp8_mb_kernel)")
show(["0 wait+above", "1 residual+fetch", "2 prediction", "3 hand-on+copy", "4 V+H edges", "5 publish", "6 emit+handdown", "7", "8 loop top", "9 band start"])
