"""The multi-GPU path shards independent images across ranks with no data-path collective; only the timing reduction
uses torch.distributed.  World-size-2 rehearsal on CPU (gloo): every rank runs the host half of the product path on its
shard (pixel-weighted split, threaded parse, native batch layout + gather) and the packed buffers it produces must be,
file by file, the ones an unsharded run produces."""
import glob
import json
import os
import subprocess
import sys
import zlib
import numpy as np
from conftest import ROOT

WORKER = r'''
import os, sys, json, glob, zlib
sys.path.insert(0, %r)
import numpy as np
import torch, torch.distributed as dist
from webp_amd import parse_many
from webp_amd.decoder import LossyBatch
from webp_amd.sharding import shard_by_weight, peek_pixels
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
paths = sorted(glob.glob(os.path.join(%r, "tests", "golden", "inputs", "[1-5].webp"))) + sorted(glob.glob(os.path.join(%r, "bench_data", "lossy_*.webp")))
datas = [open(p, "rb").read() for p in paths] * 2
weights = [peek_pixels(d) for d in datas]
lo, hi = shard_by_weight(weights, rank, world)
files = parse_many(datas[lo:hi], threads=2)
b = LossyBatch([f.frames[0] for f in files], threads=2)
sums = []
for i in range(b.n):
    d = b.desc[i]
    nmb = d.mbw * d.mbh
    nblk = (b.desc[i + 1].coeff_base if i + 1 < b.n else b.n_coeff_blocks) - d.coeff_base
    rec = b.mb[4 * d.mb_base:4 * (d.mb_base + nmb)].tobytes()
    flt = b.filt[d.mb_base:d.mb_base + nmb].tobytes()
    cof = b.coeff[16 * d.coeff_base:16 * (d.coeff_base + nblk)].tobytes()
    sums.append([lo + i, d.width, d.height, zlib.crc32(rec), zlib.crc32(flt), zlib.crc32(cof)])
gathered = [None] * world
dist.all_gather_object(gathered, {"rank": rank, "lo": lo, "hi": hi, "px": sum(weights[lo:hi]), "sums": sums})
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)    # bench.py's max-over-ranks timing
if rank == 0:
    print(json.dumps({"parts": gathered, "tmax": t.item(), "n": len(datas), "total_px": sum(weights)}))
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


def test_shard_by_weight_balances_pixels():
    from webp_amd.sharding import shard_by_weight, shard_range
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        uni = [shard_by_weight([5] * 1024, r, world) for r in range(world)]
        assert uni[0][0] == 0 and uni[-1][1] == 1024 and max(b - a for a, b in uni) - min(b - a for a, b in uni) <= 1
        w = [int(x) for x in rng.choice([46 * 32, 550 * 368, 1920 * 1080, 3840 * 2160], 300)]
        spans = [shard_by_weight(w, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == len(w) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        loads = [sum(w[a:b]) for a, b in spans]
        assert max(loads) - min(loads) <= 2 * max(w)            # within one (largest) item of each other
    assert shard_by_weight([], 0, 2) == (0, 0) and shard_by_weight([0, 0, 0], 1, 2) == (2, 3)


def test_peek_pixels_reads_the_container_headers():
    from conftest import golden_input
    from webp_amd.sharding import peek_pixels
    assert peek_pixels(golden_input("1")) == 550 * 368 and peek_pixels(golden_input("1ll")) == 400 * 301
    assert peek_pixels(golden_input("1a")) == 46 * 32 * 108 and peek_pixels(golden_input("a")) == 300 * 300
    assert peek_pixels(b"RIFF\x04\x00\x00\x00WEBP") == 0 and peek_pixels(b"junk") == 0


def _unsharded_sums(paths):
    from webp_amd import parse_many
    from webp_amd.decoder import LossyBatch
    datas = [open(p, "rb").read() for p in paths] * 2
    files = parse_many(datas, threads=2)
    b = LossyBatch([f.frames[0] for f in files])
    out = []
    for i in range(b.n):
        d = b.desc[i]
        nmb = d.mbw * d.mbh
        nblk = (b.desc[i + 1].coeff_base if i + 1 < b.n else b.n_coeff_blocks) - d.coeff_base
        out.append([i, d.width, d.height, zlib.crc32(b.mb[4 * d.mb_base:4 * (d.mb_base + nmb)].tobytes()),
                    zlib.crc32(b.filt[d.mb_base:d.mb_base + nmb].tobytes()),
                    zlib.crc32(b.coeff[16 * d.coeff_base:16 * (d.coeff_base + nblk)].tobytes())])
    return out


def test_two_rank_gloo_shards_parse_and_assemble_like_one_rank(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % (ROOT, ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    parts = sorted(res["parts"], key=lambda p: p["rank"])
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == res["n"]      # every file owned exactly once
    assert abs(parts[0]["px"] - parts[1]["px"]) <= 2 * 1280 * 720                                        # pixel-weighted: within one (largest) image
    paths = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inputs", "[1-5].webp"))) + sorted(glob.glob(os.path.join(ROOT, "bench_data", "lossy_*.webp")))
    assert parts[0]["sums"] + parts[1]["sums"] == _unsharded_sums(paths)
    assert res["tmax"] == 1.5
