"""bench.py's multi-process plumbing on CPU: `--gpus N` outside a launcher starts N ranks itself, a launcher
with another world size is refused, and rank 0's line reports how many ranks the all-reduce saw.  The
executor is replaced by a stand-in (`--selftest-cpu`, gloo); rates are not measured here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_starts_two_ranks_by_itself():
    p = _run(["--gpus", "2", "--selftest-cpu", "--workload", "tiny", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1                                   # rank 0 only
    r = lines[0]
    assert r["selftest"] and r["ok"] and r["n_gpus"] == 2 and r["ranks_seen"] == 2
    # the per-rank diagnostics of an N > 1 line: one entry per rank, gathered to rank 0, every field a number
    pr = r["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1]
    for e in pr:
        assert set(e) == {"rank", "frames", "frames_in_flight", "hw_queues", "compute_ms", "host_wait_ms", "gather_ms", "tiles"}
        assert e["frames"] == 3 and e["tiles"] == 9 and e["gather_ms"] >= 0.0 and e["compute_ms"] >= 0.0      # tiny: 144 rows = 18 tiles


def test_single_rank_runs_in_process():
    p = _run(["--selftest-cpu", "--workload", "tiny", "--steps", "1", "--warmup", "0"])
    assert p.returncode == 0, p.stderr[-2000:]
    (r,) = _json_lines(p.stdout)
    assert r["n_gpus"] == 1 and r["ranks_seen"] == 1 and r["ok"]


def test_world_size_mismatch_is_refused_without_a_result_line():
    p = _run(["--gpus", "2", "--selftest-cpu", "--workload", "tiny"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not _json_lines(p.stdout)
    assert "WORLD_SIZE=3" in p.stderr
    p = _run(["--gpus", "1", "--selftest-cpu", "--workload", "tiny"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not _json_lines(p.stdout)
