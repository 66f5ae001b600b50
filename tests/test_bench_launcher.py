"""bench.py's rank handling on CPU: ``--gpus N`` launches N ranks itself, an external launcher must agree with ``--gpus``,
and the JSON line reports the communicator's size.  ``--dry-gloo`` swaps RCCL for gloo and the GPU work for a copy."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)


def _line(stdout):
    rows = [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1, stdout
    return rows[0]


def test_self_launch_two_ranks():
    r = _run([sys.executable, BENCH, "--gpus", "2", "--dry-gloo", "--steps", "2", "--warmup", "1", "--songs-per-rank", "3"])
    assert r.returncode == 0, r.stderr[-2000:]
    row = _line(r.stdout)
    assert row["n_gpus"] == 2 and row["gather_ok"] and row["dry"]
    assert {k: row["rccl"][k] for k in ("world_size", "backend", "launcher")} == {"world_size": 2, "backend": "gloo", "launcher": "self"}
    assert row["rccl"]["gather_bytes_per_step"] == 3 * 2 * 4096 * 4


def test_double_buffered_gather_holds_the_right_step(tmp_path):
    """the REAL step() / drain() logic of the files mode (FilesPipeline: two stem buffers, one asynchronous
    gather per step, a buffer reused only after its gather drained) with a CPU stand-in engine over gloo, world 2, 6 steps:
    every gathered buffer must hold exactly the data of its own step, in order -- overlapped and blocking."""
    for extra in ([], ["--no-overlap"]):
        r = _run([sys.executable, BENCH, "--gpus", "2", "--dry-gloo", "--steps", "4", "--warmup", "2", "--songs-per-rank", "2"] + extra)
        assert r.returncode == 0, r.stderr[-2000:]
        row = _line(r.stdout)
        assert row["gather_ok"] and row["gathers_checked"] == 6 and row["gather_mismatches"] == [], row


def test_external_launcher_must_match_gpus():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", "29533", BENCH, "--dry-gloo", "--steps", "1", "--warmup", "0"]
    ok = _run(base + ["--gpus", "2"])
    assert ok.returncode == 0, ok.stderr[-2000:]
    row = _line(ok.stdout)
    assert row["n_gpus"] == 2 and row["rccl"]["launcher"] == "external"
    bad = _run(base + ["--gpus", "1"])
    assert bad.returncode != 0 and "refusing to report a mislabelled run" in (bad.stderr + bad.stdout)


def test_single_rank_line():
    r = _run([sys.executable, BENCH, "--dry-gloo", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    row = _line(r.stdout)
    assert row["n_gpus"] == 1 and row["rccl"]["world_size"] == 1
