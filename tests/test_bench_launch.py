"""bench.py's launch path on CPU: `--gpus N` without a launcher starts N ranks itself (the flag was parsed and
never read), and a launcher's WORLD_SIZE that disagrees with the flag is an error, not a silent N=1 run.
`--dry --backend gloo` = the control plane only (stream table broadcast, barriers, counter reductions), no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_gpus_2_spawns_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3", "--frames", "8"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["steps_per_rank"] == [3, 3] and out["dry"] is True
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 3 - 2 * 8 * 8160 * 3) < 1e-3 * 2 * 8 * 8160 * 3      # whole-job units / max time


def test_gpus_1_dry_is_one_rank():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "2", "--frames", "4"],
                       capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["steps_per_rank"] == [2]


def test_world_size_must_match_the_flag():
    e = _env()
    e.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry"], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode != 0 and "must agree" in r.stderr
