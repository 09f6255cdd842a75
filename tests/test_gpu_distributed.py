"""GPU side of the N>1 path on ONE device: the hash-partition kernel, and the full rank pipeline with a
world_size-1 RCCL group (partition -> all_to_all_single -> local group -> gather)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import parity_util as pu
import query_amd
from oracle import n1o
from query_amd import _ffi, plan

pytestmark = pytest.mark.gpu


def D(*names):
    return plan.field_path("default", *names)


COND = "(50 < %s)" % D("price")
KEYS = [D("cat")]
AGGS = sorted(["count(*)", "sum(%s)" % D("price"), "max(%s)" % D("user_id")])


def _device_cols(t, paths):
    import torch
    by = {c.name: c for c in t.columns}
    keep, dev = [], {}
    for p in paths:
        c = by[p]
        if c.kind == n1o.COL_DICT32:
            x = torch.from_numpy(c.codes.view(np.int32)).cuda()
            keep.append(x)
            dev[p] = (_ffi.COL_DICT32, None, None, x.data_ptr())
        else:
            a = torch.from_numpy(c.tags).cuda()
            b = torch.from_numpy(c.payload.view(np.int64)).cuda()
            keep += [a, b]
            dev[p] = (_ffi.COL_TAGGED64, a.data_ptr(), b.data_ptr(), None)
    torch.cuda.synchronize()
    return dev, keep


ARITH_COND = "(40 < (%s - %s))" % (D("price"), D("region_id"))


@pytest.mark.parametrize("jit", [0, 2], ids=["interpreter", "runtime-built"])
@pytest.mark.parametrize("keys", [KEYS, [D("price")], [D("cat"), D("region_id")], ["(%s %% 5)" % D("region_id"), D("cat")]],
                         ids=["dict", "float", "dict+int", "computed+dict"])
@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_partition_kernel_routes_every_survivor_once(nparts, keys, jit):
    """Both partition kernels — the interpreting one and the plan shape's run-time-built one (tile-sorted survivors written
    in runs, arithmetic in registers) — route every survivor exactly once, and a group key to exactly one part."""
    import torch
    n = 90_001
    t = n1o.synth_table(n, k_cat=29, zipf=True)
    cond = ARITH_COND if "%" in keys[0] else COND  # (the computed key comes with arithmetic in the Filter)
    # (two keys: aggregates over the Filter's column only, so that the shape stays within the 3 input columns of a
    #  run-time-built kernel)
    AGGS = globals()["AGGS"] if len(keys) == 1 else sorted(["count(*)", "sum(%s)" % D("price")])
    sender = query_amd.GpuFilterGroup(plan.filter_group_plan(cond, keys, AGGS))
    sender.set_option("jit", jit)
    sender.intern(list(t.dictionary))
    paths = sender.column_paths
    dev, keep = _device_cols(t, paths)
    cols = [dev[p] for p in paths]
    batch, arr = sender._make_batch(n, cols)
    cap = n
    out = (_ffi.Col * len(cols))()
    bufs = []
    for i, c in enumerate(cols):
        out[i].kind = c[0]
        if c[0] == _ffi.COL_DICT32:
            b = torch.zeros(cap * nparts, dtype=torch.int32, device="cuda")
            out[i].codes = b.data_ptr()
            bufs.append((b,))
        else:
            a = torch.zeros(cap * nparts, dtype=torch.uint8, device="cuda")
            b = torch.zeros(cap * nparts, dtype=torch.int64, device="cuda")
            out[i].tags, out[i].payload = a.data_ptr(), b.data_ptr()
            bufs.append((a, b))
    counts = torch.zeros(nparts, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    st = sender._lib.n1k_partition_device_batch(sender._h, C.byref(batch), nparts, cap, out, counts.data_ptr())
    sender._check(st)
    assert len(paths) <= 3
    assert (sender.stats()["spec_kernel"] != 0) == (jit == 2)
    cnt = counts.cpu().numpy()
    ora_sel = n1o.run(t, cond, [], [], has_group=False).selected
    assert int(cnt.sum()) == len(ora_sel)  # every survivor exactly once
    # a group key lives in exactly one part, and the union of the parts' groups is the oracle's answer
    ora = n1o.run(t, cond, keys, AGGS)
    seen = {}
    merged_keys, merged_aggs = [], []
    for d in range(nparts):
        recv = query_amd.GpuFilterGroup(plan.filter_group_plan(None, keys, AGGS))
        recv.intern(list(t.dictionary))
        rcols = []
        for p in recv.column_paths:
            i = paths.index(p)
            if cols[i][0] == _ffi.COL_DICT32:
                rcols.append((_ffi.COL_DICT32, None, None, bufs[i][0].data_ptr() + 4 * d * cap))
            else:
                rcols.append((_ffi.COL_TAGGED64, bufs[i][0].data_ptr() + d * cap, bufs[i][1].data_ptr() + 8 * d * cap, None))
        recv.process_device_items(int(cnt[d]), rcols)
        rows = recv.after_items()
        recv.done()
        for k, a in zip(rows.keys, rows.aggs):
            assert k not in seen, "group %r landed in two parts" % (k,)
            seen[k] = d
            merged_keys.append(k)
            merged_aggs.append(a)
    sender.done()
    from query_amd.gpu_operator import GroupRows
    pu.assert_same_groups(GroupRows(len(keys), len(AGGS), merged_keys, merged_aggs, []), ora, aggs=AGGS)


def test_partial_groups_export_merge_roundtrip():
    """Partial groups exported in hash-partitioned regions and merged back (as the owner ranks would after the
    all-to-all) give the single-handle answer; merging the same regions twice doubles the additive aggregates."""
    import torch
    n = 80_000
    t = n1o.synth_table(n, k_cat=300)
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "avg(%s)" % D("price"), "min(%s)" % D("price"), "max(%s)" % D("user_id")])
    op = query_amd.GpuFilterGroup(plan.filter_group_plan(COND, KEYS, aggs))
    op.intern(list(t.dictionary))
    dev, keep = _device_cols(t, op.column_paths)
    op.process_device_items(n, [dev[p] for p in op.column_paths])
    nparts, cap = 4, 256
    lib = op._lib
    region = int(lib.n1k_partial_region_bytes(op._h, cap))
    buf = torch.zeros(region * nparts, dtype=torch.uint8, device="cuda")
    op._check(lib.n1k_export_partials_device(op._h, nparts, cap, buf.data_ptr()))
    counts = [int(buf[d * region: d * region + 8].view(torch.int64).item()) for d in range(nparts)]
    assert sum(counts) == len(n1o.run(t, COND, KEYS, aggs).keys)  # every group sits in exactly one region
    with pytest.raises(query_amd.N1kError):  # too small a region is reported, not truncated
        small = torch.zeros(int(lib.n1k_partial_region_bytes(op._h, 8)) * nparts, dtype=torch.uint8, device="cuda")
        op._check(lib.n1k_export_partials_device(op._h, nparts, 8, small.data_ptr()))
    op.reopen()
    op._check(lib.n1k_merge_partials_device(op._h, nparts, cap, buf.data_ptr()))
    merged = op.after_items()
    op.done()
    ora = n1o.run(t, COND, KEYS, aggs)
    pu.assert_same_groups(merged, ora, aggs=aggs)


@pytest.mark.parametrize("mode", ["partials", "gathered"])
def test_rank_pipeline_partials_world1_rccl(mode):
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 120_000
        t = n1o.synth_table(n, k_cat=5000)  # more groups than the initial region capacity: exercises the retry
        op = qd.ShardedFilterGroup(COND, KEYS, AGGS, t.dictionary, 0, 1, 0)
        dev, keep = _device_cols(t, op.send_paths)
        raw, info = op.run_partials(n, dev) if mode == "partials" else op.run_gathered(n, dev)
        assert info["mode"] == ("partials" if mode == "partials" else "gathered partials")
        ora = n1o.run(t, COND, KEYS, AGGS)
        op.sender.sync()
        assert op.sender.stats()["rows_selected"] == ora.rows_passed and raw["ngroups"] == len(ora.keys)
        cache = {}
        from query_amd.gpu_operator import GroupRows
        got = GroupRows(1, len(AGGS), op.receiver._py_values(raw["keys"], cache), op.receiver._py_values(raw["aggs"], cache), [])
        pu.assert_same_groups(got, ora, aggs=AGGS)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("jit,subs", [(0, 1), (2, 1), (2, 2)], ids=["interpreter", "runtime-built", "runtime-built-subregions"])
def test_rank_pipeline_world1_rccl(jit, subs):
    """The row exchange end to end on one rank: both partition kernels; dense runs and (forced) per-destination sub-regions
    with their own counters, which the receiver aggregates as one segmented batch."""
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 120_000
        t = n1o.synth_table(n, k_cat=50)
        op = qd.ShardedFilterGroup(COND, KEYS, AGGS, t.dictionary, 0, 1, 0)
        op.sender.set_option("jit", jit)
        op.sender.set_option("part_subs", subs)
        op.receiver.set_option("jit", jit)  # (the segmented batch on the run-time-built scan / on the interpreter)
        if subs == 2:
            op.row_capacity = 2 * n  # (59 tiles over 8 sub-regions: an uneven deal, so generous regions)
        dev, keep = _device_cols(t, op.send_paths)
        ora = n1o.run(t, COND, KEYS, AGGS)
        for step in range(3):  # the first step sizes the regions for the whole shard, the later ones for what arrived
            raw, info = op.run_rows(n, dev)
            assert info["mode"] == "rows" and info["recv_rows"] == ora.rows_passed
            assert raw["ngroups"] == len(ora.keys)
        assert subs == 2 or max(info["region_rows"]) < n  # (regions shrank to the survivors' share)
        cache = {}
        keys = op.receiver._py_values(raw["keys"], cache)
        aggs = op.receiver._py_values(raw["aggs"], cache)
        from query_amd.gpu_operator import GroupRows
        pu.assert_same_groups(GroupRows(1, len(AGGS), keys, aggs, []), ora, aggs=AGGS)
    finally:
        dist.destroy_process_group()


def test_rank_pipeline_falls_back_to_rows_for_wide_key_values():
    """Float group keys are coded by a device-local value table, so partial groups cannot travel in packed form:
    every rank must agree (through the region headers) to use the row exchange instead."""
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 50_000
        t = n1o.synth_table(n, k_cat=50)
        keys, aggs = [D("price")], sorted(["count(*)", "max(%s)" % D("user_id")])
        op = qd.ShardedFilterGroup(COND, keys, aggs, t.dictionary, 0, 1, 0)
        dev, keep = _device_cols(t, op.send_paths)
        raw, info = op.run_partials(n, dev)
        assert info["mode"].startswith("rows")
        ora = n1o.run(t, COND, keys, aggs)
        from query_amd.gpu_operator import GroupRows
        cache = {}
        got = GroupRows(1, len(aggs), op.receiver._py_values(raw["keys"], cache), op.receiver._py_values(raw["aggs"], cache), [])
        pu.assert_same_groups(got, ora, aggs=aggs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["gathered", "partials", "rows"])
def test_rank_pipeline_grouped_tail(mode):
    """HAVING / ORDER BY / OFFSET / LIMIT across ranks: the merging handle applies them in the gathered mode; with
    hash-partitioned owners every owner keeps its first offset+limit rows and their union is ordered by
    n1k_order_rows (config 5's shape: GROUP BY cat, region_id ORDER BY SUM(price) DESC LIMIT k)."""
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 150_000
        t = n1o.synth_table(n, k_cat=300)
        keys = [D("cat"), D("region_id")]
        aggs = sorted(["sum(%s)" % D("price"), "count(*)"])
        order, limit, offset = [("sum(%s)" % D("price"), True), (D("cat"), False)], 25, 3
        having = "(5 < count(*))"
        op = qd.ShardedFilterGroup(None, keys, aggs, t.dictionary, 0, 1, 0, order=order, limit=limit, offset=offset, having=having)
        dev, keep = _device_cols(t, op.send_paths)
        raw, info = {"gathered": op.run_gathered, "partials": op.run_partials, "rows": op.run_rows}[mode](n, dev)
        k, a = raw["keys"], raw["aggs"]  # (already the union with the grouped tail applied: n1k_gather_groups)
        ora = n1o.run(t, None, keys, aggs)
        ci = aggs.index("count(*)")
        kept = [(kk, aa) for kk, aa in zip(ora.keys, ora.aggs) if aa[ci][1] > 5]
        ora.keys, ora.aggs = [x for x, _ in kept], [y for _, y in kept]
        cache = {}
        got = GroupRows(len(keys), len(aggs), op.merger._py_values(k, cache), op.merger._py_values(a, cache), [])
        pu.assert_ordered_groups(got, ora, keys, aggs, order, limit, offset)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("workload", ["config2", "config3"])
def test_row_exchange_20m_rows_subregions_world1(workload):
    """20 M rows through n1k_exchange_rows at world_size 1: enough tiles for the sub-regions (8 counters per destination)
    and the segmented receive; COUNT(DISTINCT) rides the same path (sets cannot be merged from partials).  Against the
    oracle on the same rows."""
    import torch
    import torch.distributed as dist
    from query_amd import distributed as qd
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n = 20_000_001
        t = n1o.synth_table(n, k_cat=100)
        if workload == "config2":
            cond, keys, aggs = COND, KEYS, sorted(["count(*)", "sum(%s)" % D("price")])
        else:
            cond, keys, aggs = None, KEYS, sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")])
        op = qd.ShardedFilterGroup(cond, keys, aggs, t.dictionary, 0, 1, 0)
        dev, keep = _device_cols(t, op.send_paths)
        ora = n1o.run(t, cond, keys, aggs, threads=16)
        for step in range(2):
            raw, info = op.run_rows(n, dev)
            assert info["mode"] == "rows" and info["recv_rows"] == ora.rows_passed
        assert op.sender.stats()["spec_kernel"] != 0  # the run-time-built partition kernel
        cache = {}
        gk = op.receiver._py_values(raw["keys"], cache)
        ga = op.receiver._py_values(raw["aggs"], cache)
        from query_amd.gpu_operator import GroupRows
        pu.assert_same_groups(GroupRows(1, len(aggs), gk, ga, []), ora, aggs=aggs)
    finally:
        dist.destroy_process_group()


def _run_ranks(world, fn):
    """Drive `world` ranks from their own threads (the loopback transport blocks in every collective until all ranks are in)."""
    import threading
    errs, outs = [None] * world, [None] * world

    def body(r):
        try:
            outs[r] = fn(r)
        except BaseException as e:  # noqa: BLE001 - reported below
            errs[r] = e

    ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=90)
    for e in errs:  # (a rank that died outside a collective leaves its peers waiting in the next one: its error first)
        if e is not None:
            raise e
    if any(t.is_alive() for t in ts):
        import faulthandler
        import sys
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)  # which call every rank sits in
    assert not any(t.is_alive() for t in ts), "a rank is stuck in a collective; the others returned %r" % ([o for o in outs if o is not None],)
    return outs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,jit,subs", [(2, 0, 1), (2, 2, 2), (4, 2, 2), (3, 2, 1)],
                         ids=["w2-interpreter", "w2-runtime-built-subregions", "w4-runtime-built-subregions", "w3-runtime-built-dense"])
def test_world_size_n_exchanges_over_the_loopback_transport(world, jit, subs):
    """The world_size > 1 paths of n1k_exchange_rows / n1k_exchange_partials / n1k_gather_groups on ONE GPU: the ranks are
    threads over the loopback transport (n1k_comm_create_loopback: rendezvous + device copies instead of RCCL, which
    refuses two ranks on one device).  Every rank shards one table, and every rank must end with the oracle's groups of the
    whole table — through the row exchange (region offsets per destination, header lists, sub-regions, segmented receive,
    capacities agreed through n1k_comm_max_u64), the hash-partitioned partial groups and the gathered partial groups."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    n = 240_007
    t = n1o.synth_table(n, k_cat=61, zipf=True)
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "max(%s)" % D("price")])
    ora = n1o.run(t, COND, KEYS, aggs)
    comms = qd.Comm.loopback(world, 0)
    by = {c.name: c for c in t.columns}
    shards, keep = [], []
    probe = query_amd.GpuFilterGroup(plan.filter_group_plan(COND, KEYS, aggs))
    paths = probe.column_paths
    probe.done()
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        sub = n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[lo:hi],
                                    payload=None if c.payload is None else c.payload[lo:hi],
                                    codes=None if c.codes is None else c.codes[lo:hi]) for c in t.columns], t.dictionary)
        dev, k = _device_cols(sub, paths)
        keep.append(k)
        shards.append((hi - lo, dev))

    def rank_body(r):
        op = qd.ShardedFilterGroup(COND, KEYS, aggs, t.dictionary, r, world, 0, comm=comms[r])
        for h in (op.sender, op.receiver):
            h.set_option("jit", jit)
        op.sender.set_option("part_subs", subs)
        if subs == 2:
            op.row_capacity = 2 * n  # (few tiles dealt unevenly over the sub-regions: generous regions)
        rows_n, dev = shards[r]
        got = {}
        for mode, fn in (("rows", op.run_rows), ("partials", op.run_partials), ("gathered", op.run_gathered)):
            for _ in range(2):
                raw, info = fn(rows_n, dev)
            cache = {}
            got[mode] = (GroupRows(1, len(aggs), op.merger._py_values(raw["keys"], cache), op.merger._py_values(raw["aggs"], cache), []), info)
        return got

    outs = _run_ranks(world, rank_body)
    for r in range(world):
        for mode in ("rows", "partials", "gathered"):
            rows, info = outs[r][mode]
            pu.assert_same_groups(rows, ora, aggs=aggs)
        assert outs[r]["rows"][1]["mode"] == "rows"
    # every survivor arrived at exactly one owner
    assert sum(outs[r]["rows"][1]["recv_rows"] for r in range(world)) == ora.rows_passed


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", range(int(os.environ.get("N1K_RANDOM_DIST_SEEDS", "24"))))
def test_random_plans_sharded_over_two_ranks(seed):
    """Differential test of the sharded operator: seeded random plans (conditions, arithmetic, 0-2 keys of every kind, 1-4
    aggregates incl. DISTINCT) over random mixed-type tables cut in two shards, through the partial-group exchange (which
    falls back to rows for DISTINCT and wide key values) and through the row exchange, world_size 2 over the loopback
    transport: every rank must end with the oracle's groups of the whole table."""
    import test_gpu_random_plans as rp
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    rng = np.random.default_rng(424200 + seed)
    n = int(rng.integers(2, 5000))
    t = rp.make_table(rng, n)
    cond, keys, aggs = rp.rand_plan(rng)
    if not keys:
        keys = [rp.D("s")]  # (the exchanges partition on group keys)
    try:
        probe = query_amd.GpuFilterGroup(plan.filter_group_plan(cond, keys, aggs))
    except query_amd.N1kError as e:
        if e.status == _ffi.UNSUPPORTED:
            pytest.skip("outside the device subset: " + e.message)
        raise
    paths = probe.column_paths
    probe.done()
    try:
        ora = n1o.run(t, cond, keys, aggs, threads=2)
    except n1o.OracleError as e:
        pytest.skip("outside the oracle's restated subset: %s" % e)
    world = 2
    comms = qd.Comm.loopback(world, 0)
    shards, keep = [], []
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        sub = n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[lo:hi],
                                    payload=None if c.payload is None else c.payload[lo:hi],
                                    codes=None if c.codes is None else c.codes[lo:hi]) for c in t.columns], t.dictionary)
        dev, k = _device_cols(sub, paths)
        keep.append(k)
        shards.append((hi - lo, dev))
    jit = 2 if seed % 3 == 0 else 0  # (a third of the seeds through the run-time-built kernels: ~ 3 s of compilation each)

    def rank_body(r):
        op = qd.ShardedFilterGroup(cond, keys, aggs, t.dictionary, r, world, 0, comm=comms[r])
        for h in (op.sender, op.receiver):
            h.set_option("jit", jit)
        rows_n, dev = shards[r]
        got = {}
        for mode, fn in (("partials", op.run_partials), ("rows", op.run_rows)):
            try:
                raw, info = fn(rows_n, dev)
            except query_amd.N1kError as e:
                got[mode] = e
                continue
            cache = {}
            got[mode] = GroupRows(len(keys), len(aggs), op.merger._py_values(raw["keys"], cache), op.merger._py_values(raw["aggs"], cache), [])
        return got

    old, pu.ABS_TOL = pu.ABS_TOL, 1e-9
    try:
        outs = _run_ranks(world, rank_body)
        for mode in ("partials", "rows"):
            res = [outs[r][mode] for r in range(world)]
            errs = [x for x in res if isinstance(x, Exception)]
            if errs:
                # a data error (arrays as MIN / MAX operands ...) must reach EVERY rank — nobody is left waiting
                assert len(errs) == world, "only %d of %d ranks failed: %r" % (len(errs), world, errs)
                if all(e.status in (_ffi.UNSUPPORTED, _ffi.UNSUPPORTED_DATA) for e in errs):
                    continue
                raise errs[0]
            for x in res:
                try:
                    pu.assert_same_groups(x, ora, aggs=aggs)
                except AssertionError as e:
                    raise AssertionError("%s | %s plan: %r %r %r jit %d" % (e, mode, cond, keys, aggs, jit))
    finally:
        pu.ABS_TOL = old


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_grouped_tail_and_distinct_across_ranks_over_the_loopback_transport(world):
    """Config 5's shape (GROUP BY cat, region_id HAVING ... ORDER BY SUM(price) DESC, cat LIMIT k OFFSET o) and config 3's
    (COUNT(DISTINCT user_id) + AVG(price)) at world sizes 2 and 4 on one GPU: every owner applies HAVING and keeps its first
    offset + limit rows, n1k_gather_groups orders and cuts their union; DISTINCT plans always exchange rows."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    n = 150_001
    t = n1o.synth_table(n, k_cat=300)
    keys5 = [D("cat"), D("region_id")]
    aggs5 = sorted(["sum(%s)" % D("price"), "count(*)"])
    order, limit, offset = [("sum(%s)" % D("price"), True), (D("cat"), False)], 25, 3
    having = "(5 < count(*))"
    keys3 = [D("cat")]
    aggs3 = sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")])
    comms5, comms3 = qd.Comm.loopback(world, 0), qd.Comm.loopback(world, 0)
    probe = query_amd.GpuFilterGroup(plan.filter_group_plan(None, keys5, sorted(aggs5 + aggs3)))
    paths = probe.column_paths  # (every column either plan reads)
    probe.done()
    shards, keep = [], []
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        sub = n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[lo:hi],
                                    payload=None if c.payload is None else c.payload[lo:hi],
                                    codes=None if c.codes is None else c.codes[lo:hi]) for c in t.columns], t.dictionary)
        dev, k = _device_cols(sub, paths)
        keep.append(k)
        shards.append((hi - lo, dev))

    def rank_body(r):
        rows_n, dev = shards[r]
        got = {}
        op5 = qd.ShardedFilterGroup(None, keys5, aggs5, t.dictionary, r, world, 0, order=order, limit=limit, offset=offset,
                                    having=having, comm=comms5[r])
        for mode, fn in (("gathered", op5.run_gathered), ("partials", op5.run_partials), ("rows", op5.run_rows)):
            raw, _info = fn(rows_n, dev)
            cache = {}
            got[mode] = GroupRows(len(keys5), len(aggs5), op5.merger._py_values(raw["keys"], cache), op5.merger._py_values(raw["aggs"], cache), [])
        op3 = qd.ShardedFilterGroup(None, keys3, aggs3, t.dictionary, r, world, 0, comm=comms3[r])
        assert op3.has_distinct  # (sets do not travel as partial groups: such plans exchange rows, as bench.py does)
        raw, info = op3.run_rows(rows_n, dev)
        cache = {}
        got["distinct"] = GroupRows(1, len(aggs3), op3.merger._py_values(raw["keys"], cache), op3.merger._py_values(raw["aggs"], cache), [])
        return got

    outs = _run_ranks(world, rank_body)
    ora5 = n1o.run(t, None, keys5, aggs5)
    ci = aggs5.index("count(*)")
    kept = [(kk, aa) for kk, aa in zip(ora5.keys, ora5.aggs) if aa[ci][1] > 5]
    ora5.keys, ora5.aggs = [x for x, _ in kept], [y for _, y in kept]
    ora3 = n1o.run(t, None, keys3, aggs3)
    for r in range(world):
        for mode in ("gathered", "partials", "rows"):
            pu.assert_ordered_groups(outs[r][mode], ora5, keys5, aggs5, order, limit, offset)
        pu.assert_same_groups(outs[r]["distinct"], ora3, aggs=aggs3)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["rows", "partials"])
def test_one_owners_failure_reaches_every_rank(mode):
    """A value only ONE owner's share holds makes that owner's n1k_finish fail (MAX over an array: its collation is outside
    the device subset).  Its peers would wait for it in the gather: instead the failing rank enters the collective with its
    status (n1k_gather_groups_status) and EVERY rank raises — nobody hangs, nobody returns a partial answer."""
    from query_amd import distributed as qd
    world, n = 2, 4000
    dictionary = [b"k%d" % i for i in range(8)] + [b"[1]"]
    rng = np.random.default_rng(5)
    codes = rng.integers(0, 8, n).astype(np.uint32)
    tags = np.full(n, n1o.T_INT, np.uint8)
    pay = rng.integers(0, 1000, n).astype(np.uint64)
    tags[n - 7], pay[n - 7] = n1o.T_ARRAY, 8  # one array value, in the second shard, in ONE group
    t = n1o.Table([n1o.Column(D("s"), n1o.COL_DICT32, codes=codes), n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)],
                  dictionary)
    keys, aggs = [D("s")], sorted(["count(*)", "max(%s)" % D("v")])
    comms = qd.Comm.loopback(world, 0)
    probe = query_amd.GpuFilterGroup(plan.filter_group_plan(None, keys, aggs))
    paths = probe.column_paths
    probe.done()
    shards, keep = [], []
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        sub = n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[lo:hi],
                                    payload=None if c.payload is None else c.payload[lo:hi],
                                    codes=None if c.codes is None else c.codes[lo:hi]) for c in t.columns], t.dictionary)
        dev, k = _device_cols(sub, paths)
        keep.append(k)
        shards.append((hi - lo, dev))

    def rank_body(r):
        op = qd.ShardedFilterGroup(None, keys, aggs, t.dictionary, r, world, 0, comm=comms[r])
        rows_n, dev = shards[r]
        try:
            (op.run_rows if mode == "rows" else op.run_partials)(rows_n, dev)
        except query_amd.N1kError as e:
            return e
        return None

    outs = _run_ranks(world, rank_body)
    assert all(isinstance(o, query_amd.N1kError) for o in outs), outs
    assert all(o.status == _ffi.UNSUPPORTED_DATA for o in outs), [(o.status, o.message) for o in outs]
    assert sum("peer rank" in o.message for o in outs) == world - 1  # (one owner's own error, the others were told)


def _shards(t, paths, world):
    n = t.columns[0].tags.shape[0] if t.columns[0].tags is not None else t.columns[0].codes.shape[0]
    shards, keep = [], []
    for r in range(world):
        lo, hi = n * r // world, n * (r + 1) // world
        sub = n1o.Table([n1o.Column(c.name, c.kind, tags=None if c.tags is None else c.tags[lo:hi],
                                    payload=None if c.payload is None else c.payload[lo:hi],
                                    codes=None if c.codes is None else c.codes[lo:hi]) for c in t.columns], t.dictionary)
        dev, k = _device_cols(sub, paths)
        keep.append(k)
        shards.append((hi - lo, dev))
    return shards, keep


def _paths(cond, keys, aggs):
    probe = query_amd.GpuFilterGroup(plan.filter_group_plan(cond, keys, aggs))
    paths = probe.column_paths
    probe.done()
    return paths


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["rows", "partials", "gathered"])
@pytest.mark.parametrize("site,when", [(1, 0), (1, 1), (2, 1), (3, 1), ("batch", 0), ("batch", 1)],
                         ids=["buffers-first-step", "buffers-later-step", "partition-or-export", "receiving-part", "bad-batch-first-step",
                              "bad-batch-later-step"])
def test_a_ranks_failure_around_the_collective_reaches_every_rank(mode, site, when):
    """One rank of three fails at a chosen point of ONE step (option inject_failure on its sender: 1 = the exchange's buffers
    cannot be had, 2 = its partition / export fails, 3 = its receiving part fails; or a batch whose tag pointer is null), in
    its first step (no buffers yet: it ships one scratch region to every peer) or after a good one.  Whatever the point, every
    rank returns from that step with the failing rank's status — nobody is left waiting in the all-to-all or in the gather —
    and the next step, without the fault, gives the oracle's groups on every rank (the communicator is still in step)."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    world, bad = 3, 1
    n = 60_003
    t = n1o.synth_table(n, k_cat=37)
    aggs = sorted(["count(*)", "sum(%s)" % D("price")])
    ora = n1o.run(t, COND, KEYS, aggs)
    comms = qd.Comm.loopback(world, 0)
    shards, keep = _shards(t, _paths(COND, KEYS, aggs), world)
    if site == "batch" and mode != "rows":
        pytest.skip("a batch is validated by the push of the shard in these modes: covered by the export site")

    def rank_body(r):
        op = qd.ShardedFilterGroup(COND, KEYS, aggs, t.dictionary, r, world, 0, comm=comms[r])
        fn = {"rows": op.run_rows, "partials": op.run_partials, "gathered": op.run_gathered}[mode]
        rows_n, dev = shards[r]
        seen = []
        for step in range(3):
            faulty = r == bad and step == when
            if faulty and site != "batch":
                op.sender.set_option("inject_failure", site)
            if faulty and site == "batch":
                good = op._batch(rows_n, dev)
                for i in range(len(op.send_paths)):
                    if good[0].cols[i].kind == _ffi.COL_TAGGED64:
                        saved, good[0].cols[i].tags = good[0].cols[i].tags, None
                        which = i
                        break
            try:
                raw, _info = fn(rows_n, dev)
                cache = {}
                seen.append(GroupRows(1, len(aggs), op.merger._py_values(raw["keys"], cache), op.merger._py_values(raw["aggs"], cache), []))
            except query_amd.N1kError as e:
                seen.append(e)
            if faulty and site == "batch":
                good[0].cols[which].tags = saved
        return seen

    outs = _run_ranks(world, rank_body)
    want = {1: _ffi.OOM, 2: _ffi.DEVICE_ERROR, 3: _ffi.DEVICE_ERROR, "batch": _ffi.INVALID}[site]
    for r in range(world):
        for step in range(3):
            got = outs[r][step]
            if step == when and site == 3 and mode == "gathered" and r != bad:
                # (every rank merged every rank's groups and no gather follows: the receiving part is each rank's own)
                assert not isinstance(got, Exception), (r, step, got)
                pu.assert_same_groups(got, ora, aggs=aggs)
            elif step == when:
                assert isinstance(got, query_amd.N1kError), "rank %d returned groups from the faulty step" % r
                assert got.status == want, (r, got.status, got.message)
            else:
                assert not isinstance(got, Exception), (r, step, got)
                pu.assert_same_groups(got, ora, aggs=aggs)
    if site == 3:  # the receiving part is one rank's own: its peers were told in the gather (gathered mode has none)
        told = sum(isinstance(outs[r][when], Exception) and "peer rank" in outs[r][when].message for r in range(world))
        assert told == (0 if mode == "gathered" else world - 1), [o[when].message for o in outs]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", ["array-under-less-than", "float-key-without-wide-values"])
def test_what_a_senders_partition_finds_reaches_every_rank(case):
    """The row exchange's SENDER side sees the data first: a Filter that has to order an array, a group key value that does
    not pack (a float key with wide_values = 0).  Its partition kernel drops such rows and raises its own error flags —
    which now travel in the verdict word of every region it ships: every rank's step fails with N1K_UNSUPPORTED_DATA, as the
    single-GPU path does for the same data, instead of returning groups that miss rows."""
    from query_amd import distributed as qd
    world, n = 2, 6000
    rng = np.random.default_rng(11)
    dictionary = [b"k%d" % i for i in range(8)] + [b"[1]", b"[2]"]
    codes = rng.integers(0, 8, n).astype(np.uint32)
    tags = np.full(n, n1o.T_INT, np.uint8)
    pay = rng.integers(0, 1000, n).astype(np.uint64)
    opts = {}
    if case == "array-under-less-than":
        tags[n - 5], pay[n - 5] = n1o.T_ARRAY, 8  # one array, in the second shard
        # (an array against an array: only then does `<` have to order arrays, which the device cannot)
        cond, keys = "(%s < %s)" % (D("v"), D("w")), [D("s")]
        wt, wp = tags.copy(), (pay + 1).astype(np.uint64)
        wp[n - 5] = 9  # (another array: equal codes are equal values and need no ordering)
        cols = [n1o.Column(D("s"), n1o.COL_DICT32, codes=codes), n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay),
                n1o.Column(D("w"), n1o.COL_TAGGED64, tags=wt, payload=wp)]
    else:
        ftags = np.full(n, n1o.T_FLOAT, np.uint8)
        fpay = (rng.integers(0, 50, n) + 0.5).astype(np.float64).view(np.uint64)
        cond, keys = None, [D("f")]
        cols = [n1o.Column(D("f"), n1o.COL_TAGGED64, tags=ftags, payload=fpay), n1o.Column(D("v"), n1o.COL_TAGGED64, tags=tags, payload=pay)]
        opts = {"wide_values": 0}
    aggs = sorted(["count(*)", "sum(%s)" % D("v")])
    t = n1o.Table(cols, dictionary)
    comms = qd.Comm.loopback(world, 0)
    shards, keep = _shards(t, _paths(cond, keys, aggs), world)

    def rank_body(r):
        op = qd.ShardedFilterGroup(cond, keys, aggs, t.dictionary, r, world, 0, comm=comms[r])
        for k, v in opts.items():
            for h in (op.sender, op.receiver):
                h.set_option(k, v)
        rows_n, dev = shards[r]
        try:
            op.run_rows(rows_n, dev)
        except query_amd.N1kError as e:
            return e
        return None

    outs = _run_ranks(world, rank_body)
    assert all(isinstance(o, query_amd.N1kError) for o in outs), outs
    assert all(o.status == _ffi.UNSUPPORTED_DATA for o in outs), [(o.status, o.message) for o in outs]


@pytest.mark.timeout(300)
def test_several_exchanges_per_step_on_one_communicator():
    """processItem* then afterItems: a rank sends its shard in several batches, one n1k_exchange_rows each, before the
    owner's n1k_finish.  Every exchange reuses the communicator's two buffers: it waits (on the device) for the owner's scans
    of the previous one.  World size 2 over the loopback transport, three batches per rank, against the oracle."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    world, nb = 2, 3
    n = 300_000
    t = n1o.synth_table(n, k_cat=53)
    aggs = sorted(["count(*)", "sum(%s)" % D("price"), "count(distinct %s)" % D("user_id")])
    ora = n1o.run(t, COND, KEYS, aggs)
    comms = qd.Comm.loopback(world, 0)
    paths = _paths(COND, KEYS, aggs)
    pieces, keep = _shards(t, paths, world * nb)

    def rank_body(r):
        op = qd.ShardedFilterGroup(COND, KEYS, aggs, t.dictionary, r, world, 0, comm=comms[r])
        lib = op.sender._lib
        op.receiver.reopen()
        op.sender.reopen()
        for b in range(nb):
            rows_n, dev = pieces[r * nb + b]
            batch = op.sender.make_device_batch(rows_n, [dev[p] for p in op.send_paths])
            op.sender._check(lib.n1k_exchange_rows(op.comm._h, op.sender._h, C.byref(batch[0]), op.receiver._h, n))
        raw = op.receiver.after_items_raw()
        local = _ffi.Result()
        keys_a, aggs_a = np.ascontiguousarray(raw["keys"]), np.ascontiguousarray(raw["aggs"])
        local.ngroups, local.nkeys, local.naggs = raw["ngroups"], raw["nkeys"], raw["naggs"]
        local.keys = C.cast(keys_a.ctypes.data, C.POINTER(_ffi.Value)) if keys_a.size else None
        local.aggs = C.cast(aggs_a.ctypes.data, C.POINTER(_ffi.Value)) if aggs_a.size else None
        out = _ffi.Result()
        op.merger._check(lib.n1k_gather_groups(op.comm._h, op.merger._h, C.byref(local), C.byref(out)))
        res = op._result_dict(out)
        cache = {}
        return GroupRows(1, len(aggs), op.merger._py_values(res["keys"], cache), op.merger._py_values(res["aggs"], cache), [])

    for rows in _run_ranks(world, rank_body):
        pu.assert_same_groups(rows, ora, aggs=aggs)


@pytest.mark.timeout(900)
def test_world_size_8_configs_4_and_5_over_the_loopback_transport():
    """BASELINE configs 4 and 5 are 8-rank configurations: their own queries at world size 8 on one GPU (loopback transport).
    Config 4 = config 2's and config 3's queries with rows hash-partitioned on cat by the all-to-all (the row exchange; for
    config 2 also both partial-group modes); config 5 = GROUP BY cat, region_id ORDER BY SUM(price) DESC LIMIT 100 (row
    exchange and hash-partitioned partial groups, each owner's top rows gathered and cut).  Every rank must end with the
    oracle's answer over the whole table, and every survivor must reach exactly one owner."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    world = 8
    n = 400_003
    t = n1o.synth_table(n, k_cat=1000)
    aggs2 = ["sum(%s)" % D("price")]
    aggs3 = sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")])
    keys5 = [D("cat"), D("region_id")]
    order5, limit5 = [("sum(%s)" % D("price"), True)], 100
    paths = _paths(COND, keys5, sorted(aggs2 + aggs3))
    shards, keep = _shards(t, paths, world)
    comms = {k: qd.Comm.loopback(world, 0) for k in ("c2", "c3", "c5")}

    def rank_body(r):
        rows_n, dev = shards[r]
        got = {}

        def rows_of(op, raw, nk, na):
            cache = {}
            return GroupRows(nk, na, op.merger._py_values(raw["keys"], cache), op.merger._py_values(raw["aggs"], cache), [])

        op2 = qd.ShardedFilterGroup(COND, KEYS, aggs2, t.dictionary, r, world, 0, comm=comms["c2"][r])
        for mode, fn in (("rows", op2.run_rows), ("partials", op2.run_partials), ("gathered", op2.run_gathered)):
            raw, info = fn(rows_n, dev)
            got["c2-" + mode] = (rows_of(op2, raw, 1, 1), info)
        op3 = qd.ShardedFilterGroup(None, KEYS, aggs3, t.dictionary, r, world, 0, comm=comms["c3"][r])
        raw, info = op3.run_rows(rows_n, dev)
        got["c3-rows"] = (rows_of(op3, raw, 1, 2), info)
        op5 = qd.ShardedFilterGroup(None, keys5, aggs2, t.dictionary, r, world, 0, order=order5, limit=limit5, comm=comms["c5"][r])
        for mode, fn in (("rows", op5.run_rows), ("partials", op5.run_partials)):
            raw, info = fn(rows_n, dev)
            got["c5-" + mode] = (rows_of(op5, raw, 2, 1), info)
        return got

    outs = _run_ranks(world, rank_body)
    ora2 = n1o.run(t, COND, KEYS, aggs2, threads=4)
    ora3 = n1o.run(t, None, KEYS, aggs3, threads=4)
    ora5 = n1o.run(t, None, keys5, aggs2, threads=4)
    for r in range(world):
        for mode in ("rows", "partials", "gathered"):
            pu.assert_same_groups(outs[r]["c2-" + mode][0], ora2, aggs=aggs2)
        pu.assert_same_groups(outs[r]["c3-rows"][0], ora3, aggs=aggs3)
        for mode in ("rows", "partials"):
            pu.assert_ordered_groups(outs[r]["c5-" + mode][0], ora5, keys5, aggs2, order5, limit5, None)
    assert sum(outs[r]["c2-rows"][1]["recv_rows"] for r in range(world)) == ora2.rows_passed
    assert sum(outs[r]["c3-rows"][1]["recv_rows"] for r in range(world)) == n
    assert sum(outs[r]["c5-rows"][1]["recv_rows"] for r in range(world)) == n


@pytest.mark.timeout(900)
def test_skewed_keys_size_each_owners_regions_on_their_own():
    """SURVEY.md 8e, skew: under Zipf(1.0) keys the owner of the hottest key receives about twice its share.  Every destination
    has its own region capacity (n1k_rows_step_v, agreed per destination after the first step from what the senders really
    wrote): the hot owner's regions grow, the other seven stay at their own size — what a rank ships in all stays within 1.3 x
    of the uniform case, where ONE capacity for all 64 regions (sized by the largest owner) shipped about 1.9 x.  World size 8
    over the loopback transport, config 2's and config 3's queries, against the oracle."""
    from query_amd import distributed as qd
    from query_amd.gpu_operator import GroupRows
    world, n = 8, 800_000
    aggs2 = sorted(["count(*)", "sum(%s)" % D("price")])
    aggs3 = sorted(["count(distinct %s)" % D("user_id"), "avg(%s)" % D("price")])
    paths = _paths(COND, KEYS, sorted(aggs2 + aggs3))
    shipped = {}
    for zipf in (False, True):
        t = n1o.synth_table(n, k_cat=1000, zipf=zipf)
        ora2 = n1o.run(t, COND, KEYS, aggs2, threads=4)
        ora3 = n1o.run(t, None, KEYS, aggs3, threads=4)
        shards, keep = _shards(t, paths, world)
        comms2, comms3 = qd.Comm.loopback(world, 0), qd.Comm.loopback(world, 0)

        def rank_body(r):
            rows_n, dev = shards[r]
            got = {}
            for name, cond, aggs, comms in (("c2", COND, aggs2, comms2), ("c3", None, aggs3, comms3)):
                op = qd.ShardedFilterGroup(cond, KEYS, aggs, t.dictionary, r, world, 0, comm=comms[r])
                for _ in range(3):  # the first step sizes all regions alike; from the second on every destination has its own
                    raw, info = op.run_rows(rows_n, dev)
                cache = {}
                got[name] = (GroupRows(1, len(aggs), op.merger._py_values(raw["keys"], cache), op.merger._py_values(raw["aggs"], cache), []), info)
            return got

        outs = _run_ranks(world, rank_body)
        for r in range(world):
            pu.assert_same_groups(outs[r]["c2"][0], ora2, aggs=aggs2)
            pu.assert_same_groups(outs[r]["c3"][0], ora3, aggs=aggs3)
            assert outs[r]["c2"][1]["region_rows"] == outs[0]["c2"][1]["region_rows"]  # one vector, agreed by all ranks
        assert sum(outs[r]["c2"][1]["recv_rows"] for r in range(world)) == ora2.rows_passed
        shipped[zipf] = {k: outs[0][k][1]["region_rows"] for k in ("c2", "c3")}
    for k in ("c2", "c3"):
        uni, zpf = shipped[False][k], shipped[True][k]
        assert max(uni) <= 1.25 * min(uni), uni                      # uniform keys: every owner alike
        assert max(zpf) >= 1.5 * min(zpf), zpf                       # Zipf: the hot owner's regions are the large ones ...
        assert sum(zpf) <= 1.3 * sum(uni), (sum(zpf), sum(uni))      # ... and only they: a rank ships about what it shipped before
