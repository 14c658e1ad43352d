"""A/B timing of library builds on the bench workload: for every library given (default: the in-tree one) a child process
times vp8_mb_kernel alone (mixed batch; q75-only and q90-only batches with --streams), the finishing pass alone and a
pipelined step, and checks 21 frames of the pipelined run against the oracle.
  python tools/mb_time.py [--streams] [--steps 30] [lib.so ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(streams, steps):
    import torch
    import bench
    from webp_amd.device import LossyPipeline

    def t(fn, reps=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {}
    files, batch, db, paths, _ = bench.build_batch(1024, "lossy1080_*.webp", "cuda:0")
    out["mb_alone"] = round(t(db.recon_filter), 3)
    out["finish_alone"] = round(t(db.finish), 3)
    pipe = LossyPipeline("cuda:0", one_launch=True)
    for k in range(2):
        db.workspace(k)
    for _ in range(3):
        pipe.submit(db)
    pipe.drain(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(db)
    pipe.drain(); torch.cuda.synchronize()
    out["pipelined"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
    try:
        out["verified"] = bench.verify_outputs(db, batch, paths, 1024)["count"]
    except SystemExit as e:
        out["verified"] = "MISMATCH: " + str(e)[:80]
    del db, batch, files, pipe
    torch.cuda.empty_cache()
    if streams:
        for name, pat in (("q75", "lossy1080_q75_0.webp"), ("q90", "lossy1080_q90_0.webp"), ("simple", "lossy1080_q75_simple_1.webp")):
            files, batch, db, paths, _ = bench.build_batch(1024, pat, "cuda:0")
            out["mb_" + name] = round(t(db.recon_filter, 5), 3)
            del db, batch, files
            torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child("--streams" in sys.argv, int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30)
        sys.exit(0)
    libs = [a for a in sys.argv[1:] if a.endswith(".so")] or [None]
    flags = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for lib in libs:
        env = dict(os.environ)
        if lib:
            env["WEBP_AMD_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + flags, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print("%-44s %s" % (os.path.basename(lib) if lib else "in-tree", line[-1] if line else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])), flush=True)
