"""Stand-in for bench.py's per-rank half, used by tests/test_bench_launcher.py through WEBP_BENCH_RANK_SCRIPT: joins a gloo
process group like a real rank would join RCCL, and rank 0 prints one JSON line.  WEBP_STUB_FAIL_RANK=R in the environment makes rank R exit 3."""
import argparse
import json
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--warmup", type=int, default=0)
ap.add_argument("--total-images", type=int, default=0)
ap.add_argument("--pattern", default="")
args = ap.parse_args()

import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dist.barrier()
dist.destroy_process_group()
if rank == int(os.environ.get("WEBP_STUB_FAIL_RANK", "-1")):
    sys.exit(3)
if rank == 0:
    print("some banner on stdout that is not the JSON line")
    print(json.dumps({"n_gpus": world, "gpus_arg": args.gpus, "total_images": args.total_images, "pattern": args.pattern}), flush=True)
