#!/usr/bin/env python3
"""Benchmark of the hot path: batched 1080p lossy WebP, recon + loop filter + YUV->ARGB on MI355X.

One "step" = one pass of the device pipeline (reconstruction + loop filter as one kernel, then the finishing pass,
producing both the filtered YUV420 planes and ARGB) over a batch of synthetic encoded 1920x1080 VP8 images whose packed
macroblock / coefficient / filter buffers are already resident in HBM (the host entropy decode and the H2D copy happen
once, before the timed region).  Successive steps are software-pipelined by the library itself (wp_vp8_pipeline_* of the C
ABI, csrc/pipeline.cpp: two HIP streams, rotating workspaces): the memory-bound finishing pass of step k runs beside the
issue-bound kernel of step k+1; every step does all of its work and the timed region ends with both streams drained.  `--serial` runs the steps on one stream
instead (each kernel alone on the GPU); `--alone` adds the single-stream figure to the pipelined run's JSON line.
Metric: decoded Mpix/s, counting visible pixels, whole job over all GPUs.

  python bench.py --gpus N --steps K --warmup W [--images 1024]
  N > 1: one rank per GPU; the batch is sharded by image index with no data-path collective (weak scaling: every rank
  decodes its own `--images` images; `--total-images M` shards ONE batch of M images instead: cfg 5 is
  `--gpus 8 --total-images 8192 --pattern lossy2160_q75_0.webp`).  Under torch.distributed.run (WORLD_SIZE set) this
  process is one rank and --gpus must equal the world size; started plainly with --gpus N > 1 it launches the N ranks
  itself as child processes (launch_ranks) and relays rank 0's line.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def build_batch(n_images, pattern, device):
    from webp_amd import parse
    from webp_amd.decoder import LossyBatch
    from webp_amd.device import DeviceLossyBatch
    paths = sorted(glob.glob(os.path.join(ROOT, "bench_data", pattern)))
    assert paths, "bench_data/%s missing" % pattern
    t0 = time.time()
    files = [parse(open(p, "rb").read()) for p in paths]
    t_parse = (time.time() - t0) / len(files)
    frames = [files[i % len(files)].frames[0] for i in range(n_images)]
    batch = LossyBatch(frames)
    return files, batch, DeviceLossyBatch(batch, device), paths, t_parse


def cpu_baseline(paths, budget_s=12.0):
    """Oracle (CPU restatement of the reference, 1 thread) on a bounded sample of the same files:
    WebPDecoder.decode + Tools.toArgb per image."""
    from oracle import oracle
    datas = [open(p, "rb").read() for p in paths]
    oracle.decode(datas[0])            # warm-up
    t0 = time.time()
    n = px = 0
    while time.time() - t0 < budget_s:
        img = oracle.decode(datas[n % len(datas)])
        px += img["width"] * img["height"]
        n += 1
    dt = time.time() - t0
    out = {"value": px / dt / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": "%d x 1920x1080 lossy (bench_data/lossy1080_*.webp cycled), decode+toArgb, %.1f s, "
                     "reference Haxe toolchain unavailable on the box: timed the C restatement (oracle/)" % (n, dt)}
    # The reference has no threading; "all host cores" = independent single-threaded processes on the same files
    # (SURVEY 8d).  Child processes, so that nothing here forks a process that has initialised the GPU.
    import subprocess
    cores = min(16, os.cpu_count() or 1)             # the GPU box gives one GPU's job 16 cores
    code = ("import sys,time,glob,os\n"
            "sys.path.insert(0, %r)\n"
            "from oracle import oracle\n"
            "ds=[open(p,'rb').read() for p in %r]\n"
            "oracle.decode(ds[0]); t0=time.time(); n=px=0\n"
            "while time.time()-t0 < %f:\n"
            "    im=oracle.decode(ds[n %% len(ds)]); px+=im['width']*im['height']; n+=1\n"
            "print(px/(time.time()-t0)/1e6)\n") % (ROOT, list(paths), budget_s * 0.6)
    try:
        procs = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
        rates = [float(pr.communicate(timeout=budget_s * 3 + 60)[0].strip().splitlines()[-1]) for pr in procs]
        out["all_cores"] = {"value": sum(rates), "unit": "Mpix/s", "cores": cores, "how": "%d independent single-threaded processes, %.0f s each" % (cores, budget_s * 0.6)}
    except Exception as e:                            # the one-core figure stands on its own
        out["all_cores"] = {"value": None, "error": str(e)[:200]}
    return out


def verify_outputs(db, batch, paths, n_images):
    """After the timed region: the outputs the last step left in HBM (planes + ARGB) of a sample of frames -- frame 0, the
    last frame and a run of indices in the batch's second half, i.e. frames that are a persistent workgroup's second or
    later frame, covering every distinct stream -- against the CPU oracle's decode of the same file.  Returns the number
    of frames compared; raises on the first difference (a bench line is only printed for bit-exact output)."""
    import numpy as np
    from oracle import oracle
    n = n_images
    half = n // 2
    idx = sorted(set([0, n - 1, max(0, half - 1)] + [min(n - 1, half + k) for k in range(len(paths) + 2)] +
                     [min(n - 1, half + (n - half) * k // 6) for k in range(1, 6)]))
    want = {}
    for i in idx:
        s = i % len(paths)
        if s not in want:
            want[s] = oracle.decode(open(paths[s], "rb").read())
        w = want[s]
        y, u, v = db.planes(i)
        if not (np.array_equal(y, w["y"]) and np.array_equal(u, w["u"]) and np.array_equal(v, w["v"])):
            raise SystemExit("bench.py: YUV planes of frame %d (%s) differ from the oracle" % (i, os.path.basename(paths[s])))
        if not np.array_equal(db.argb_of(i), w["to_argb"]):
            raise SystemExit("bench.py: ARGB of frame %d (%s) differs from the oracle" % (i, os.path.basename(paths[s])))
    return {"count": len(idx), "indices": idx, "against": "oracle.decode (planes + ARGB, bit-exact)"}


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the launcher.  It
    starts N rank processes (one per GPU) as CHILDREN through torch.distributed.run -- before anything here has touched the
    GPU, and never by replacing this process -- relays rank 0's JSON line and returns the job's exit status (non-zero if
    any rank failed).  WEBP_BENCH_RANK_SCRIPT names the per-rank program (default: this file; tests/ substitutes a stub so
    that the launcher can be exercised without GPUs)."""
    import socket
    import subprocess
    script = os.environ.get("WEBP_BENCH_RANK_SCRIPT") or os.path.abspath(__file__)
    if not os.environ.get("WEBP_BENCH_RANK_SCRIPT"):
        import torch                                  # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n_gpus:
            sys.stderr.write("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing to run (a smaller run would be "
                             "reported under the wrong n_gpus)\n" % (n_gpus, have))
            return 2
    with socket.socket() as s:                        # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank job failed (exit status %d%s)\n" % (n_gpus, proc.returncode, "" if lines else ", no JSON line from rank 0"))
        return proc.returncode or 1
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=1024, help="images per GPU (BASELINE.json configs[1]: 1024; 4096 = 8 frames per persistent "
                    "workgroup gives every wavefront the same number of bands and is ~10 %% faster per image)")
    ap.add_argument("--pattern", default="lossy1080_*.webp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the records of the other BASELINE.json configurations (lossless, animation + alpha, 4K, end to end)")
    ap.add_argument("--total-images", type=int, default=0, help="size of ONE global batch that is sharded over the ranks by webp_amd.sharding (cfg 5: 8192 with --pattern lossy2160_q75_0.webp); "
                    "0 = every rank decodes --images images (weak scaling, the default)")
    ap.add_argument("--serial", action="store_true", help="one stream per step, no overlap between steps (kernel times as when each runs alone)")
    ap.add_argument("--split", type=float, default=0.6, help="share of a step's frames whose finishing pass runs beside the next reconstruction (the rest: beside the next loop filter)")
    ap.add_argument("--finish-delay-us", type=float, default=None, help="least time a finishing pass is held back behind the launch of the persistent kernel beside it (the library pipeline waits for that kernel's workgroups to be placed anyway; default 0; 40 for --py-pipeline, which only has the time)")
    ap.add_argument("--two-phase", action="store_true", help="round-1 schedule: reconstruction and loop filter as two kernels (R tiles through HBM), the finishing pass of the step before split over both")
    ap.add_argument("--py-pipeline", action="store_true", help="pipeline the steps with webp_amd.device.LossyPipeline (torch streams, torch.cuda._sleep as hold-off) instead of the library's own wp_vp8_pipeline_* (the default)")
    ap.add_argument("--workspaces", type=int, default=2, help="tile workspaces the pipelined steps rotate through")
    ap.add_argument("--alone", action="store_true", help="after the timed region also time the single-stream fused call (config.serial_ms_per_step)")
    args = ap.parse_args()
    args.one_launch = not args.two_phase             # default: vp8_mb_kernel (reconstruction + loop filter, one kernel) with the finishing pass of the step before beside it

    # --gpus N is the number of ranks of the job.  Under a launcher (WORLD_SIZE set) it must agree with it; without one,
    # N > 1 makes this process the launcher of N child ranks (nothing has touched the GPU yet).
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        if int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: refusing to report one as the other"
                             % (args.gpus, os.environ["WORLD_SIZE"]))
    elif args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    dist = None
    saved_stdout = None
    if world > 1 or os.environ.get("WEBP_BENCH_FORCE_DIST"):      # the env switch exercises the RCCL path on a 1-GPU box
        # RCCL prints a version banner on stdout when NCCL_DEBUG=VERSION is set: keep stdout for the one JSON line by
        # pointing file descriptor 1 at stderr until that line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(device))

    shard = None
    if args.total_images > 0:                        # the global batch's frames all have one size here: equal weights
        from webp_amd.sharding import shard_by_weight
        lo, hi = shard_by_weight([1] * args.total_images, rank, world)
        shard = (lo, hi)
        args.images = hi - lo
    files, batch, db, paths, t_parse = build_batch(args.images, args.pattern, device)
    n_mb = batch.n_mb
    px_per_step = sum(batch.desc[i].width * batch.desc[i].height for i in range(batch.n))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from webp_amd.device import LossyPipeline, NativePipeline
    if args.serial:
        from webp_amd._lib import lib as _wl
        _wl().wp_hint_finish_overlap(0)            # the finishing pass runs on its own here
    native = not args.serial and args.one_launch and not args.py_pipeline
    if args.serial:
        pipe = None
    elif native:                                     # the C ABI's pipeline: its own streams, workspaces (allocated here) and hold-off
        pipe = NativePipeline(device, work_bytes=db.work_bytes, workspaces=args.workspaces, timing=True, hold_us=args.finish_delay_us)
    else:
        pipe = LossyPipeline(device, split=args.split, finish_delay_us=40.0 if args.finish_delay_us is None else args.finish_delay_us, one_launch=args.one_launch, workspaces=args.workspaces)
        for k in range(args.workspaces):
            db.workspace(k)                          # allocated here, not inside the timed region

    marks = []                                       # per step: timing events around each kernel, on its launch stream

    def run(n_steps, timing=None):
        for _ in range(n_steps):
            if native:
                pipe.submit(db, after_current_stream=False)
            elif pipe is not None:
                pipe.submit(db, timing=timing)
            elif timing is None:
                db.decode_to_argb()
            elif args.one_launch:                    # one kernel for reconstruction + loop filter, then the finishing pass
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(); db.recon_filter(); ev[1].record(); db.finish(); ev[2].record()
                timing.append((ev[0], ev[1], ev[1], ev[1], ev[1], ev[2]))
            else:                                    # three launches, staged so that events fit between them
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record(); db.recon(planes=False); ev[1].record(); db.loopfilter(planes=False); ev[2].record(); db.finish(); ev[3].record()
                timing.append((ev[0], ev[1], ev[1], ev[2], ev[2], ev[3]))
        if pipe is not None:
            pipe.drain()

    run(max(args.warmup, 1) if pipe is not None else args.warmup)      # the streams' first use is slow: never time it
    barrier()
    if native:
        pipe.stats(reset=True)
    t0 = time.perf_counter()
    run(args.steps, marks)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * elapsed / args.steps]
    total_px = px_per_step * world
    if dist is not None:
        t = torch.tensor([elapsed, -elapsed, float(px_per_step)], device=device, dtype=torch.float64)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        rank_ms = [1e3 * float(-tmax[1].item()) / args.steps, 1e3 * float(tmax[0].item()) / args.steps]      # fastest, slowest rank
        elapsed = float(tmax[0].item())
        total_px = float(tsum[2].item())             # pixels all ranks decoded per step (ranks of a sharded batch can differ by an image)
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_px * args.steps / elapsed / 1e6

    # ---- per-kernel launch durations over the timed region (HIP events on each kernel's launch stream) and the roofline
    #      of the dominant kernel.  In the pipelined run a reconstruction shares the GPU with the previous step's
    #      finishing pass, so its duration here is longer than when it runs alone (`--serial`, profiles/*serial*). ----
    acc = [0.0, 0.0, 0.0]
    finish_launches = 0
    halves = [0.0, 0.0, 0]                          # finishing pass beside reconstruction / beside the loop filter
    for m in marks:
        if isinstance(m, dict):                      # pipelined: the finishing pass of a step is two launches (one, for the last step)
            acc[0] += m["r0"].elapsed_time(m["r1"])
            acc[1] += m["f0"].elapsed_time(m["f1"])
            if "ca" in m:
                ha, hb = m["c0"].elapsed_time(m["ca"]), m["cb"].elapsed_time(m["c1"])
                acc[2] += ha + hb
                halves[0] += ha; halves[1] += hb; halves[2] += 1
                finish_launches += 2
            else:
                acc[2] += m["c0"].elapsed_time(m["c1"])
                finish_launches += 1
        else:
            for k in range(3):
                acc[k] += m[2 * k].elapsed_time(m[2 * k + 1])
            finish_launches += 1
    if native:            # HIP timing events the pipeline keeps around both launches of every step, each on the stream its kernel runs on
        ms_mb, ms_fin, n_done = pipe.stats()
        assert n_done == args.steps, (n_done, args.steps)
        t_recon, t_filter, t_conv = ms_mb / n_done, ms_mb / n_done, ms_fin / n_done
        finish_per_step = 1.0
    else:
        t_recon, t_filter, t_conv = (a / len(marks) for a in acc)          # per step
        finish_per_step = finish_launches / len(marks)
    serial_ms = None
    if args.alone and pipe is not None:              # the same steps on one stream (no overlap), for reference
        db.decode_to_argb(); torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            db.decode_to_argb()
        torch.cuda.synchronize()
        serial_ms = 1e3 * (time.perf_counter() - ts0) / args.steps
    verified = verify_outputs(db, batch, paths, args.images)          # every rank checks its own outputs (a mismatch ends the job)
    in_bytes = batch.input_bytes()
    yuv_bytes = 384 * n_mb
    argb_bytes = 4 * px_per_step
    alg = {"vp8_recon_kernel": in_bytes - batch.filt.nbytes + yuv_bytes,
           "vp8_filter_kernel": batch.filt.nbytes + 2 * yuv_bytes,
           "vp8_finish_kernel": 2 * yuv_bytes + argb_bytes}       # tiles in, row-major planes + ARGB out
    times = {"vp8_recon_kernel": t_recon, "vp8_filter_kernel": t_filter, "vp8_finish_kernel": t_conv}
    if args.one_launch:         # reconstruction + loop filter are one kernel: packed inputs in, S tiles out
        alg = {"vp8_mb_kernel": in_bytes + yuv_bytes, "vp8_finish_kernel": alg["vp8_finish_kernel"]}
        times = {"vp8_mb_kernel": t_recon, "vp8_finish_kernel": t_conv}
    dom = max(times, key=times.get)
    # The two kernels of a pipelined step run beside each other for the whole step and are balanced to end together, so which
    # of them is a few microseconds longer changes from run to run: within 3 % the persistent kernel (the one bound by
    # instruction issue, with the lower fraction of the HBM roofline) is the one reported.
    if args.one_launch and times["vp8_mb_kernel"] >= 0.97 * times[dom]:
        dom = "vp8_mb_kernel"
    achieved = alg[dom] / (times[dom] * 1e-3) / 1e9                    # bytes per step / time per step = bytes per launch / time per launch
    launches = finish_per_step if dom == "vp8_finish_kernel" else 1.0
    # HBM traffic of that kernel from the PMC counters: measured by a separate rocprofv3 --pmc run of this same command
    # (tools/pmc.sh + tools/pmc_traffic.list), committed under profiles/; scaled if --images differs from that run.
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic.json" if args.one_launch else "r01_traffic.json")))
        key = next(k for k in tj["kernels"] if dom in k)            # rocprofv3 spells template instances "void vp8_mb_kernel<8>"
        traffic = int(tj["kernels"][key]["hbm_bytes"] * args.images / tj["images_per_launch"] / launches)
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "alg_bytes_per_launch": int(alg[dom] / launches), "avg_launch_ms": round(times[dom] / launches, 4),
                "launches_per_step": round(launches, 3),
                "stages_ms": {k: round(v, 4) for k, v in times.items()},
                "stages_GBs": {k: round(alg[k] / (times[k] * 1e-3) / 1e9, 2) for k in times},
                "finish_halves_ms": [round(halves[0] / halves[2], 4), round(halves[1] / halves[2], 4)] if halves[2] else None,
                "pipeline_alg_bytes_per_px": round((in_bytes + yuv_bytes + argb_bytes) / px_per_step, 3),
                "pipeline_frac": round((in_bytes + yuv_bytes + argb_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

    # measured device-to-device copy bandwidth on this GPU (read + write bytes), the practical HBM ceiling beside the nominal peak
    nbytes = min(db.argb.numel() // 2, 1 << 31)
    src_t, dst_t = db.argb[:nbytes], db.argb[nbytes:2 * nbytes]
    dst_t.copy_(src_t); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst_t.copy_(src_t)
    e1.record(); e1.synchronize()
    roofline["measured_copy_GBs"] = round(5 * 2 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)

    line = {"metric": "decoded Mpix/s (whole node) for batched 1080p lossy WebP", "value": round(value, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if args.total_images > 0 else "weak", "vs_baseline": None, "dtype": "i32/u8",
            "data": "synthetic",
            "config": {"workload": "%d x %dx%d VP8 lossy (%s) per GPU: recon + loop filter + finishing pass (YUV420 planes + ARGB out), inputs resident in HBM"
                                   % (args.images, batch.desc[0].width, batch.desc[0].height,
                                      "libwebp: 8 x q75 m4 normal filter + q90 + q50 sharp3 + simple-filter variant" if args.pattern == "lossy1080_*.webp"
                                      else "bench_data/" + args.pattern),
                       "images_per_gpu": args.images, "distinct_streams": len(paths),
                       "sharding": "images across ranks, no collective" + ("" if shard is None else "; one global batch of %d split by webp_amd.sharding.shard_by_weight, rank 0 holds [%d, %d)" % (args.total_images, shard[0], shard[1])),
                       "rank_ms_per_step_min_max": [round(v, 4) for v in (rank_ms if len(rank_ms) == 2 else rank_ms * 2)],
                       "macroblocks_per_gpu": n_mb, "coeff_blocks_per_mb": round(batch.n_coeff_blocks / max(n_mb, 1), 3),
                       "host_parse_ms_per_image": round(1e3 * t_parse, 2),
                       "step_pipelining": "none (--serial)" if pipe is None else
                                          (("wp_vp8_pipeline_* (C ABI): " if native else "webp_amd.device.LossyPipeline: ") + "2 HIP streams: finishing pass of step k beside the reconstruction + loop filter kernel of step k+1; %d workspaces alternate" % args.workspaces if args.one_launch else
                                           "2 HIP streams, 2 phases: finishing pass of step k (first %d %% of the frames) beside reconstruction of step k+1, "
                                           "the rest beside its loop filter; workspaces alternate" % round(100 * args.split)),
                       "serial_ms_per_step": None if serial_ms is None else round(serial_ms, 4)},
            "verified_frames": None if verified is None else verified["count"], "verified": verified,
            "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_secondary:
        del db, batch, files, pipe
        import gc
        gc.collect(); torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_secondary
        line["secondary"] = bench_secondary.run_all(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(paths)
    if dist is not None:
        dist.destroy_process_group()
    if saved_stdout is not None:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
