"""N>1 path on CPU: the exchange protocol of query_amd/distributed.py (counts all-to-all, one all_to_all_single
per column buffer, gather of the finished groups) under gloo with world_size 2.  The per-rank compute steps
(filter+partition, local group) are played by numpy + the CPU oracle here; on GPUs they are libn1k.so kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORLD = 2
ROWS_PER_RANK = 30_000
K_CAT = 23


def D(*names):
    s = "`default`"
    for n in names:
        s = "(%s.`%s`)" % (s, n)
    return s


COND = "(50 < %s)" % D("price")
KEYS = [D("cat")]
AGGS = sorted(["count(*)", "sum(%s)" % D("user_id"), "max(%s)" % D("price"), "count(%s)" % D("price")])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, result_path):
    from oracle import n1o
    from query_amd import distributed as qd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    total = ROWS_PER_RANK * WORLD
    shard = n1o.synth_table(ROWS_PER_RANK, k_cat=K_CAT, first_row=rank * ROWS_PER_RANK, total_rows=total)
    by = {c.name: c for c in shard.columns}
    # 1. filter + hash partition on the group key (CPU stand-in for n1k_partition_device_batch)
    sel = n1o.run(shard, COND, [], [], has_group=False).selected.astype(np.int64)
    cat = by[D("cat")].codes[sel]
    dest = (cat.astype(np.int64) * 2654435761 >> 7) % WORLD
    order = np.argsort(dest, kind="stable")
    sel = sel[order]
    counts = torch.from_numpy(np.bincount(dest, minlength=WORLD).astype(np.int64))
    send_cols = [torch.from_numpy(by[D("cat")].codes[sel].astype(np.int32)),
                 torch.from_numpy(by[D("price")].tags[sel].copy()), torch.from_numpy(by[D("price")].payload[sel].view(np.int64).copy()),
                 torch.from_numpy(by[D("user_id")].tags[sel].copy()), torch.from_numpy(by[D("user_id")].payload[sel].view(np.int64).copy())]
    # 2. the exchange under test
    recv_counts = qd.exchange_counts(counts)
    recv = qd.exchange_rows(send_cols, counts.tolist(), recv_counts.tolist())
    assert all(len(r) == int(recv_counts.sum()) for r in recv)
    # 3. local InitialGroup..FinalGroup on the owned rows (CPU stand-in: the oracle, no Filter)
    local = n1o.Table([n1o.Column(D("cat"), n1o.COL_DICT32, codes=recv[0].numpy().view(np.uint32)),
                       n1o.Column(D("price"), n1o.COL_TAGGED64, tags=recv[1].numpy(), payload=recv[2].numpy().view(np.uint64)),
                       n1o.Column(D("user_id"), n1o.COL_TAGGED64, tags=recv[3].numpy(), payload=recv[4].numpy().view(np.uint64))],
                      shard.dictionary)
    res = n1o.run(local, None, KEYS, AGGS)
    # every group must live on exactly one rank
    rec = np.zeros((len(res.keys), 8 * (1 + 2 * len(AGGS))), dtype=np.uint8)
    flat = rec.view(np.int64)
    for g, (k, a) in enumerate(zip(res.keys, res.aggs)):
        flat[g, 0] = shard.dictionary.index(k[0][1])
        for i, (t, v) in enumerate(a):
            flat[g, 1 + 2 * i] = t
            if t == n1o.T_FLOAT:
                flat[g, 2 + 2 * i] = np.float64(v).view(np.int64)
            elif t == n1o.T_STRING:
                flat[g, 2 + 2 * i] = shard.dictionary.index(v)
            else:
                flat[g, 2 + 2 * i] = v if v is not None else 0
    # 4. gather on rank 0
    allg = qd.gather_groups(rec, dst=0)
    fixed = qd.FixedGather(rec.shape[1], capacity=4)(rec, torch.device("cpu"))  # too small on purpose: must retry
    if rank == 0:
        assert fixed.shape == allg.shape and np.array_equal(np.sort(fixed.view(np.int64)[:, 0]), np.sort(allg.view(np.int64)[:, 0]))
    # ranks that own very different numbers of groups (more than a default slot on one of them, next to none on the
    # other): the slot size is agreed on across the ranks, never derived from the local count
    mine = np.arange((5000 if rank == 1 else 3) * 24, dtype=np.uint8).reshape(-1, 24) + np.uint8(rank)
    both = qd.FixedGather(24)(mine, torch.device("cpu"))
    assert both.shape == (5003, 24) and np.array_equal(both[:3] if rank == 0 else both[:3], (np.arange(3 * 24, dtype=np.uint8).reshape(-1, 24)))
    assert np.array_equal(both[3:], np.arange(5000 * 24, dtype=np.uint8).reshape(-1, 24) + np.uint8(1))
    if rank == 0:
        np.save(result_path, allg.view(np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_hash_partitioned_exchange_world2(tmp_path):
    from oracle import n1o
    n1o.build()
    port = _free_port()
    out = str(tmp_path / "groups.npy")
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = np.load(out)
    total = ROWS_PER_RANK * WORLD
    full = n1o.synth_table(total, k_cat=K_CAT)
    ora = n1o.run(full, COND, KEYS, AGGS)
    want = {}
    for k, a in zip(ora.keys, ora.aggs):
        want[full.dictionary.index(k[0][1])] = a
    assert got.shape[0] == len(want), "a group was split across ranks or lost"
    seen = set()
    for row in got:
        code = int(row[0])
        assert code not in seen
        seen.add(code)
        for i, (t, v) in enumerate(want[code]):
            assert int(row[1 + 2 * i]) == t
            if t == n1o.T_FLOAT:
                assert np.int64(row[2 + 2 * i]).view(np.float64) == pytest.approx(v, rel=1e-9)
            elif t == n1o.T_STRING:
                assert int(row[2 + 2 * i]) == full.dictionary.index(v)
            else:
                assert int(row[2 + 2 * i]) == (v if v is not None else 0)
