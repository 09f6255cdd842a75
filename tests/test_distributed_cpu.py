"""N > 1 on CPU (no GPU in this container): what the product does on the HOST side of a multi-rank run.

The exchange itself lives in libn1k.so (RCCL / loopback transport) and needs a device: tests/test_gpu_distributed.py drives it
at world sizes 1-8 on one GPU.  Here, with two real processes over gloo:
  * `python bench.py --gpus 2` without a launcher's environment starts its two ranks itself (before anything touches torch or
    HIP), they rendezvous, rank 0's communicator id reaches rank 1, the shard bounds of the synthetic data set add up, and
    rank 0's JSON line is relayed (--dry-run: everything but the device work);
  * without a GPU the same command fails on EVERY rank with a clear message and a non-zero exit — it never prints a
    one-GPU line under an N-GPU flag;
  * under an external launcher (RANK / WORLD_SIZE set, as the driver's torch.distributed.run does) the ranks do not
    launch again;
  * shard_bounds: weak and strong scaling, shards that differ by one row.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_bench_starts_its_own_ranks_world_2_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--total-rows", "1000001"], env=_clean_env(),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] and out["ranks_agree"]
    assert out["rows_per_rank"] == [500000, 500001] and out["scaling"] == "strong"


@pytest.mark.timeout(300)
def test_bench_under_an_external_launcher_does_not_launch_again():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry-run", "--rows", "1000"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1000:] for o in outs]
    line0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(line0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]  # only rank 0 prints
    out = json.loads(line0[0])
    assert out["rows_per_rank"] == [1000, 1000] and out["scaling"] == "weak" and out["total_rows"] == 2000


@pytest.mark.timeout(300)
def test_bench_n_gpus_without_gpus_fails_on_every_rank():
    import query_amd
    if query_amd.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu", "--steps", "1", "--warmup", "0"], env=_clean_env(),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout  # no result line of any kind
    assert "ranks failed" in r.stderr and "(0, " in r.stderr and "(1, " in r.stderr  # both ranks stopped, none was left waiting
    assert "GPU" in r.stderr


def test_gpus_flag_must_match_the_launchers_world_size():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_shard_bounds():
    from types import SimpleNamespace
    from query_amd import distributed as qd
    for world in (1, 2, 3, 8):
        a = SimpleNamespace(total_rows=0, rows=1000)
        got = [qd.shard_bounds(a, r, world) for r in range(world)]
        assert all(g[0] == 1000 * world and g[2] == 1000 for g in got) and [g[1] for g in got] == [1000 * r for r in range(world)]
        a = SimpleNamespace(total_rows=100_000_001, rows=7)
        got = [qd.shard_bounds(a, r, world) for r in range(world)]
        assert sum(g[2] for g in got) == 100_000_001 and max(g[2] for g in got) - min(g[2] for g in got) <= 1
        assert all(got[r][1] + got[r][2] == (got[r + 1][1] if r + 1 < world else 100_000_001) for r in range(world))
