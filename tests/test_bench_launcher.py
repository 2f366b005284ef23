"""bench.py --gpus N must really run N ranks: the launcher, the rendezvous on 127.0.0.1, the ranks_seen all-reduce and
the JSON line are exercised here on CPU with the gloo backend (--dry-run: no GPU work); the reference's counterpart is train.py:14-31
(mp.spawn + init_process_group)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _line(out: str) -> dict:
    rows = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(rows) == 1, out
    return json.loads(rows[0])


def test_gpus_2_launches_two_ranks_without_torchrun():
    r = _run(["--gpus", "2", "--tiny", "--backend", "gloo", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["dry_run"] is True


def test_strong_scaling_flag_travels_to_the_ranks():
    r = _run(["--gpus", "2", "--tiny", "--backend", "gloo", "--dry-run", "--scaling", "strong"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"


def test_world_size_that_contradicts_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--tiny", "--backend", "gloo", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_single_rank_dry_run():
    r = _run(["--tiny", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1
