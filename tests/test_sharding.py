"""The multi-GPU path shards independent images across ranks with no data-path collective; only the
timing reduction uses torch.distributed.  World-size-2 rehearsal on CPU (gloo)."""
import os
import subprocess
import sys
import numpy as np
from conftest import ROOT

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from webp_amd.sharding import shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 11
lo, hi = shard_range(n, rank, world)
mine = torch.zeros(n, dtype=torch.int64); mine[lo:hi] = 1
dist.all_reduce(mine)                       # every image owned exactly once
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)    # bench.py's max-over-ranks timing
if rank == 0:
    print(json.dumps({"owned": mine.tolist(), "tmax": t.item(), "lo": lo, "hi": hi}))
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    from webp_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo(tmp_path):
    import json
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["owned"] == [1] * 11 and res["tmax"] == 1.5
