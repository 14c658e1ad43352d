"""bench.py --gpus N: with N > 1 and no launcher around it, bench.py starts the N ranks itself (child processes through
torch.distributed.run); under a launcher the world size must agree with --gpus.  CPU only: the per-rank program is a stub
(tests/launcher_stub.py) joined over gloo, so what is tested is the launcher, its argument relay and its exit status."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
STUB = os.path.join(ROOT, "tests", "launcher_stub.py")


def run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_gpus_2_starts_two_ranks_and_relays_rank0_line():
    r = run(["--gpus", "2", "--total-images", "8192", "--pattern", "lossy2160_q75_0.webp"], WEBP_BENCH_RANK_SCRIPT=STUB)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                 # exactly ONE line on stdout: the JSON
    line = json.loads(lines[0])
    assert line == {"n_gpus": 2, "gpus_arg": 2, "total_images": 8192, "pattern": "lossy2160_q75_0.webp"}


def test_a_failing_rank_fails_the_job():
    r = run(["--gpus", "2"], WEBP_BENCH_RANK_SCRIPT=STUB, WEBP_STUB_FAIL_RANK="1")
    assert r.returncode != 0
    assert "2-rank job failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_2_without_gpus_refuses_instead_of_running_one_rank():
    """On a box without (enough) GPUs `--gpus 2` must not degrade into a 1-GPU result."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has two GPUs")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert "--gpus 2 asked for" in r.stderr and "refusing" in r.stderr
    assert r.stdout.strip() == ""


def test_world_size_must_agree_with_gpus():
    r = run(["--gpus", "8"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr and "--gpus 8" in r.stderr
