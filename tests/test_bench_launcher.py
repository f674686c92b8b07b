"""`python bench.py --gpus N` without torch.distributed.run starts its own ranks: N fresh child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before anything touches a GPU.  Driven here with --dry-ranks
(every rank prints its environment and exits), so no GPU is needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _run(args, env=None, timeout=300):
    return subprocess.run([sys.executable, BENCH] + args, env=env or _env(), capture_output=True, text=True,
                          timeout=timeout)


def _records(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and '"rank"' in line:
            out.append(json.loads(line))
    return out


def test_self_launch_starts_every_rank():
    p = _run(["--gpus", "3", "--dry-ranks", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr
    # rank 0 owns stdout (where the JSON line of a real run goes); the other ranks' output goes to stderr
    first = _records(p.stdout)
    assert [r["rank"] for r in first] == [0]
    rest = _records(p.stderr)
    assert sorted(r["rank"] for r in rest) == [1, 2]
    recs = first + rest
    assert all(r["world"] == 3 and r["local_rank"] == r["rank"] for r in recs)
    masters = {r["master"] for r in recs}
    assert len(masters) == 1 and next(iter(masters)).startswith("127.0.0.1:")
    assert int(next(iter(masters)).split(":")[1]) > 0


def test_launcher_route_still_works_under_torchrun_env():
    """What torch.distributed.run sets: the script is one rank and must not start more."""
    env = _env()
    env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    p = _run(["--gpus", "2", "--dry-ranks"], env=env)
    assert p.returncode == 0, p.stderr
    recs = _records(p.stdout)
    assert recs == [{"rank": 1, "local_rank": 1, "world": 2, "master": "127.0.0.1:29512"}]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
    p = _run(["--gpus", "2", "--dry-ranks"], env=env)
    assert p.returncode != 0
    assert "WORLD_SIZE=4" in p.stderr


def test_failed_rank_fails_the_job():
    """A rank that exits non-zero makes the parent exit non-zero (here: every rank refuses an unknown dtype)."""
    p = _run(["--gpus", "2", "--dry-ranks", "--dtype", "fp8"])
    assert p.returncode != 0


def test_single_gpu_dry_run_prints_one_record():
    p = _run(["--gpus", "1", "--dry-ranks"])
    assert p.returncode == 0, p.stderr
    assert _records(p.stdout) == [{"rank": 0, "local_rank": 0, "world": 1, "master": ":"}]
