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


def test_world_8_files_mode():
    """First contact with eight ranks must be boring (VERDICT r5 #7a): the exact `bench.py --gpus 8 --mode files` code path -- self launch of
    eight ranks, rendezvous on 127.0.0.1, FilesPipeline's double-buffered gather of every rank's stems to rank 0 -- over gloo on the CPU: one JSON
    line, world size 8, every gathered buffer holding its own step's data of its own rank."""
    r = _run([sys.executable, BENCH, "--gpus", "8", "--dry-gloo", "--mode", "files", "--steps", "3", "--warmup", "1", "--songs-per-rank", "2",
              "--master-port", "29541"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    row = _line(r.stdout)
    assert row["n_gpus"] == 8 and row["dry"] and row["mode"] == "files" and row["gather_ok"] and row["gathers_checked"] == 4 and row["gather_mismatches"] == []
    assert {k: row["rccl"][k] for k in ("world_size", "backend", "launcher")} == {"world_size": 8, "backend": "gloo", "launcher": "self"}
    assert row["rccl"]["gather_bytes_per_step"] == 7 * 2 * 2 * 4096 * 4          # seven foreign ranks' stems (2 songs x [2, 4096] floats each)
    # the line's structure is the one-rank line's (same code, world 1)
    one = _line(_run([sys.executable, BENCH, "--dry-gloo", "--mode", "files", "--steps", "3", "--warmup", "1", "--songs-per-rank", "2"]).stdout)
    assert set(one) == set(row) and set(one["rccl"]) == set(row["rccl"]) and one["rccl"]["world_size"] == 1


def test_world_8_chunks_mode():
    """`bench.py --gpus 8 --mode chunks` (strong scaling of ONE song: 55 chunks in ranges of 7, 7, ..., 6, seam chunks to the right neighbour, local
    fold, one gather of [2, N / 8] slabs) over gloo with the stand-in engine: the folded song equals the input on rank 0 while every chunk slot a
    rank does not hold is NaN-poisoned."""
    r = _run([sys.executable, BENCH, "--gpus", "8", "--dry-gloo", "--mode", "chunks", "--steps", "2", "--warmup", "1", "--master-port", "29543"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    row = _line(r.stdout)
    assert row["n_gpus"] == 8 and row["dry"] and row["mode"] == "chunks" and row["fold_ok"], row
    assert row["rccl"]["world_size"] == 8 and row["rccl"]["backend"] == "gloo" and row["scaling"] == "strong"
    assert [tuple(x) for x in row["chunk_ranges"]] == [(0, 7), (7, 14), (14, 21), (21, 28), (28, 35), (35, 42), (42, 49), (49, 55)]
    assert row["demix_calls_rank0"] == 3                                         # one range per call, three calls
    one = _line(_run([sys.executable, BENCH, "--dry-gloo", "--mode", "chunks", "--steps", "2", "--warmup", "1"]).stdout)
    assert set(one) == set(row) and one["fold_ok"] and one["rccl"]["world_size"] == 1
