"""bench.py --gpus N without a launcher starts N ranks itself (before any GPU call); rehearsed here with --dry-run (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_flag_spawns_that_many_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["dry_run"] is True


def test_gpus_flag_must_match_the_launcher():
    e = _env()
    e.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=e, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
