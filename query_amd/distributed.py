"""Multi-GPU Filter -> Group -> Aggregate: one process per GPU; the collectives run inside libn1k.so (RCCL over xGMI).

The reference has no distributed query path (README.md:61-65); inside one process it fans Parallel copies into
one serial IntermediateGroup through an in-memory queue (execution/exchange.go:161-251).  Here every rank holds a row
shard and the ranks meet in ONE exchange step, behind the C ABI (include/n1k.h, "multi-GPU"):

  * n1k_exchange_rows      Filter + hash partition of the survivors on the group key into one packed region per
                           destination, ONE all-to-all (counts, verdicts and every column's rows travel together),
                           InitialGroup .. FinalGroup on the owner: a group lives on exactly one rank, COUNT(DISTINCT)
                           needs no set merge;
  * n1k_exchange_partials  every rank aggregates its shard first and only the partial groups travel — all-gathered
                           (every rank merges all of them and holds the result) or hash-partitioned to owners;
  * n1k_gather_groups(_status)  the owners' finished groups to every rank, then the plan's ORDER BY / LIMIT over the
                           union; a rank whose own step failed enters it with its status, so that every rank raises.
  * n1k_comm_create_loopback   the same exchange with the ranks as THREADS of one process (Comm.loopback): world sizes
                           > 1 on a single GPU (tests/test_gpu_distributed.py).

This module is the thin Python caller (bench.py, tests): torch.distributed only carries the communicator id at start-up
and the bench's barriers — no row and no group crosses it.  One step is ONE call through the ABI (n1k_rows_step /
n1k_partials_step), which also owns the failure rules (include/n1k.h, "Failures"): a failure that every rank sees alike
(n1k_failure_is_global) ends the step without a gather, a rank's own failure travels in the gather.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class Comm:
    """n1k_comm: one per rank.  The id travels through torch.distributed (bootstrap only)."""

    def __init__(self, rank: int, world: int, device: int):
        import torch
        import torch.distributed as dist
        from query_amd import _ffi
        self._lib = _ffi.lib()
        dev = torch.device("cuda", device)
        ident = torch.zeros(_ffi.COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * _ffi.COMM_ID_BYTES)()
            st = self._lib.n1k_comm_unique_id(buf)
            if st != 0:
                raise RuntimeError("n1k_comm_unique_id failed: %d" % st)
            ident = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
        ident = ident.to(dev)
        if world > 1:
            dist.broadcast(ident, src=0)
        raw = bytes(ident.cpu().numpy().tobytes())
        self._h = C.c_void_p()
        st = self._lib.n1k_comm_create(raw, rank, world, device, C.byref(self._h))
        if st != 0:
            raise RuntimeError("n1k_comm_create failed: %s" % (self._lib.n1k_comm_last_error(None) or b"").decode())
        self.rank, self.world = rank, world

    @classmethod
    def loopback(cls, world: int, device: int) -> "List[Comm]":
        """n1k_comm_create_loopback: `world` communicators whose ranks are threads of this process on one device (tests of
        the world_size > 1 paths on a single GPU: drive every rank from its own thread)."""
        from query_amd import _ffi
        lib = _ffi.lib()
        arr = (C.c_void_p * world)()
        st = lib.n1k_comm_create_loopback(world, device, arr)
        if st != 0:
            raise RuntimeError("n1k_comm_create_loopback failed: %s" % (lib.n1k_comm_last_error(None) or b"").decode())
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c._lib, c._h, c.rank, c.world = lib, C.c_void_p(arr[r]), r, world
            out.append(c)
        return out

    def done(self):
        if self._h:
            self._lib.n1k_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.done()
        except Exception:
            pass


class ShardedFilterGroup:
    """One rank's share of the distributed operator: a sender handle (Filter + InitialGroup over the shard), a receiver
    handle (the same grouping without the Filter: it merges partial groups or aggregates received rows, applies HAVING
    and keeps its first offset+limit rows), and a merger handle that carries the exact grouped tail over the union."""

    def __init__(self, condition: Optional[str], keys: Sequence[str], aggs: Sequence[str], dictionary: Sequence[bytes],
                 rank: int, world: int, device: int, order=None, limit=None, offset=None, having=None, comm: Optional[Comm] = None,
                 **options):
        import query_amd
        from query_amd import plan
        self.rank, self.world, self.device = rank, world, device
        self.comm = comm if comm is not None else Comm(rank, world, device)
        self.tail = order is not None or limit is not None or offset is not None or having is not None
        self.sender = query_amd.GpuFilterGroup(plan.filter_group_plan(condition, keys, aggs), device=device)
        owner_limit = None if limit is None else int(limit) + int(offset or 0)
        self.receiver = query_amd.GpuFilterGroup(
            plan.filter_group_plan(None, keys, aggs, order=order, limit=owner_limit, having=having) if self.tail
            else plan.filter_group_plan(None, keys, aggs), device=device, **options)
        self.merger = query_amd.GpuFilterGroup(
            plan.filter_group_plan(None, keys, aggs, order=order, limit=limit, offset=offset, having=having),
            device=device, **options) if self.tail else self.receiver
        for h in {id(self.sender): self.sender, id(self.receiver): self.receiver, id(self.merger): self.merger}.values():
            h.intern(list(dictionary))
        self.send_paths = self.sender.column_paths
        self.has_distinct = any("(distinct " in a.lower() for a in aggs)
        self.partial_capacity = None  # groups per region: agreed on in the first step
        self.row_capacity = None      # rows per region
        self._first_row_capacity = None
        self._caps, self._caps_agreed = None, False  # per-destination capacities agreed after the first step
        self._sbatch = None

    GATHER_LIMIT = 32 << 20  # partial groups of ALL ranks within this many bytes per rank: all-gather them

    def _batch(self, nrows: int, cols_by_path: Dict[str, tuple]):
        if self._sbatch is None or self._sbatch[0] is not cols_by_path or self._sbatch[1] != nrows:
            self._sbatch = (cols_by_path, nrows, self.sender.make_device_batch(nrows, [cols_by_path[p] for p in self.send_paths]))
        return self._sbatch[2]

    def _max(self, handle, value: int) -> int:
        out = C.c_uint64()
        handle._check(handle._lib.n1k_comm_max_u64(self.comm._h, handle._h, int(value), C.byref(out)))
        return int(out.value)

    # ------------------------------------------------------------------ exchange of partial groups
    def run_partials(self, nrows: int, cols_by_path: Dict[str, tuple], replicate: bool = False) -> Tuple[dict, dict]:
        """n1k_partials_step: every rank aggregates its shard (the single-GPU kernels), ONE collective moves the partial groups —
        an all-gather (replicate and few groups) or an all-to-all of hash-partitioned regions — the receiver merges, finishes,
        and (hash-partitioned) the finished groups are gathered.  A region that overflows voids the step on every rank alike:
        all of them retry with 4 x the capacity; wide key values (coded per handle) send all of them to the row exchange."""
        from query_amd import _ffi
        from query_amd.gpu_operator import N1kError
        snd, lib = self.sender, self.sender._lib
        batch = self._batch(nrows, cols_by_path)
        if self.partial_capacity is None:
            # first step only: size the regions from the groups the shards really hold
            snd.reopen()
            snd.process_device_batch(batch)
            snd.sync()
            ng = self._max(snd, int(snd.stats()["groups_out"]))
            full = 1 << max(10, int(np.ceil(np.log2(ng * 1.25 + 64))))
            if replicate and int(lib.n1k_partial_region_bytes(snd._h, full)) * self.world <= self.GATHER_LIMIT:
                self.partial_capacity = full
            else:
                self.partial_capacity = 1 << max(10, int(np.ceil(np.log2(ng / self.world * 1.5 + 64))))
        out, worst = _ffi.Result(), C.c_int(0)
        while True:
            cap = self.partial_capacity
            region = int(lib.n1k_partial_region_bytes(snd._h, cap))
            gathered = replicate and region * self.world <= self.GATHER_LIMIT
            rcv = self.merger if gathered else self.receiver
            st = int(lib.n1k_partials_step(self.comm._h, snd._h, C.byref(batch[0]), rcv._h, self.merger._h, cap, 1 if gathered else 0,
                                           C.byref(out), C.byref(worst)))
            everywhere = bool(lib.n1k_failure_is_global(rcv._h)) or bool(lib.n1k_failure_is_global(snd._h))
            if st == _ffi.REGION_FULL:
                self.partial_capacity *= 4  # (a region overflowed: on every rank alike, no gather took place)
                continue
            if st == _ffi.UNSUPPORTED and everywhere:  # wide key values are coded per handle: such groups travel as rows
                raw, info = self.run_rows(nrows, cols_by_path)
                info["mode"] = "rows (wide key values)"
                return raw, info
            break
        self._raise(st, worst.value, (rcv, snd, self.merger))
        stats = snd.stats()
        info = {"mode": "gathered partials" if gathered else "partials", "region_bytes": region,
                "scan_ms": float(stats["device_ms"]), "spec_kernel": int(stats["spec_kernel"])}
        return self._result_dict(out), info

    def _raise(self, st: int, worst: int, handles):
        from query_amd.gpu_operator import N1kError
        if st != 0:  # this rank's own failure (its peers learnt it in the headers or in the gather)
            msg = b""
            for hnd in handles:
                msg = msg or (hnd._lib.n1k_last_error(hnd._h) or b"")
            raise N1kError(st, msg.decode(errors="replace"))
        if worst != 0:
            raise N1kError(int(worst), "a peer rank's step failed with status %d" % worst)

    def run_gathered(self, nrows: int, cols_by_path: Dict[str, tuple]) -> Tuple[dict, dict]:
        return self.run_partials(nrows, cols_by_path, replicate=True)

    # ------------------------------------------------------------------ exchange of rows
    def run_rows(self, nrows: int, cols_by_path: Dict[str, tuple]) -> Tuple[dict, dict]:
        """n1k_rows_step_v: Filter + hash partition + ONE all-to-all + InitialGroup on the owner, then the gather.  The first step
        ships regions sized for the whole shard; from then on every DESTINATION has its own capacity: what the sender that sends
        it most really wrote for it (+ 10 %) — under skewed keys the hot owner's regions grow, the other world - 1 stay small."""
        from query_amd import _ffi
        snd, rcv, lib = self.sender, self.receiver, self.sender._lib
        batch = self._batch(nrows, cols_by_path)
        W = self.world
        if self.row_capacity is None and self._caps is None and self._first_row_capacity is None:
            # the regions for one destination have ONE size on every rank: the first step sizes them from the LARGEST shard
            # (shards differ by a row under strong scaling; found by the loopback test — RCCL would have hung on it)
            self._first_row_capacity = max(4096, int(self._max(snd, nrows) * 1.1 / W) + 4096)
        out, worst = _ffi.Result(), C.c_int(0)
        fixed = self.row_capacity is not None  # (a caller's own capacity — an int or one per destination — is kept as it is)
        while True:
            if fixed:
                caps = [int(self.row_capacity)] * W if isinstance(self.row_capacity, (int, np.integer)) else [int(x) for x in self.row_capacity]
            else:
                caps = self._caps if self._caps is not None else [self._first_row_capacity] * W
            arr = (C.c_uint64 * W)(*caps)
            # n1k_rows_step_v: both resets, the exchange, the owner's n1k_finish and the gather in ONE call through the ABI
            st = int(lib.n1k_rows_step_v(self.comm._h, snd._h, C.byref(batch[0]), rcv._h, self.merger._h, arr, C.byref(out), C.byref(worst)))
            if st == _ffi.REGION_FULL:  # (a region overflowed: on every rank alike, no gather took place)
                if fixed:
                    self.row_capacity = [c * 2 for c in caps]
                else:
                    self._caps = [c * 2 for c in caps]
                continue
            break
        if not fixed and not self._caps_agreed and st == _ffi.OK and worst.value == 0:  # (a failed step is known as such on every rank)
            sent = (C.c_uint64 * W)()
            snd._check(lib.n1k_exchange_sent_rows(self.comm._h, snd._h, sent))
            most = (C.c_uint64 * W)()
            snd._check(lib.n1k_comm_max_u64_v(self.comm._h, snd._h, W, sent, most))
            slack = 1.1 if W > 1 else 1.02  # (shards are not identical; an overflow doubles the capacities)
            self._caps = [min(caps[d], max(4096, int(int(most[d]) * slack) + 4096)) for d in range(W)]
            self._caps_agreed = True
        self._raise(st, worst.value, (rcv, snd, self.merger))
        stats = snd.stats()
        return self._result_dict(out), {"mode": "rows", "region_rows": list(caps), "scan_ms": float(stats["device_ms"]),
                                        "recv_rows": int(rcv.stats()["rows_selected"])}

    @staticmethod
    def _result_dict(out) -> dict:
        from query_amd.gpu_operator import GpuFilterGroup
        dt = GpuFilterGroup._VALUE_DT
        n, nk, na = int(out.ngroups), int(out.nkeys), int(out.naggs)

        def arr(ptr, count):
            if not count or not ptr:
                return np.zeros(0, dtype=dt)
            return np.frombuffer(bytearray(C.string_at(ptr, count * dt.itemsize)), dtype=dt)

        return {"ngroups": n, "nkeys": nk, "naggs": na,
                "keys": arr(out.keys, n * nk).reshape(n, nk) if nk else np.zeros((n, 0), dt),
                "aggs": arr(out.aggs, n * na).reshape(n, na) if na else np.zeros((n, 0), dt)}

    def done(self):
        for h in (self.sender, self.receiver, self.merger):
            h.done()
        self.comm.done()


def shard_bounds(args, rank: int, world: int) -> Tuple[int, int, int]:
    """(rows in all, this rank's first row, its row count): weak scaling by default (every rank owns --rows rows of one global
    synthetic data set), --total-rows T = strong scaling (T rows in all, split evenly; shards differ by at most one row)."""
    total_rows = args.total_rows if args.total_rows else args.rows * world
    first = total_rows * rank // world
    return total_rows, first, total_rows * (rank + 1) // world - first


def bench_main(args, rank: int, world: int, local_rank: int):
    """bench.py --gpus N (N > 1, or --force-dist): the hot path sharded over N ranks.  Weak scaling by default (every rank
    owns --rows rows of one global synthetic data set); --total-rows T = strong scaling (T rows in all: config 4 is
    100 M, config 5 is 1 B over 8 GPUs)."""
    import torch
    import torch.distributed as dist
    import bench
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # bootstrap + the bench's barriers only
    wl = bench.workloads()[args.workload]
    strong = bool(args.total_rows)
    total_rows, first, rows = shard_bounds(args, rank, world)
    cols = bench.DeviceColumns(rows, args.kcat, bool(args.zipf), first, total_rows, local_rank)
    op = ShardedFilterGroup(wl["cond"], wl["keys"], wl["aggs"], bench.synth_dictionary(args.kcat), rank, world,
                            local_rank, order=wl.get("order"), limit=wl.get("limit"))
    for o in getattr(args, "opt", []):  # engine options (tuning experiments): every handle of the rank
        k, v = o.split("=")
        for hnd in {id(x): x for x in (op.sender, op.receiver, op.merger)}.values():
            hnd.set_option(k, int(v))
    dev = torch.device("cuda", local_rank)
    mode = args.exchange
    if mode == "auto":  # the configuration north_star names: rows hash-partitioned on the group key by one all-to-all
        mode = "rows"

    def step():
        if op.has_distinct or mode == "rows":
            return op.run_rows(rows, cols.by_path)
        if mode == "partials":
            return op.run_partials(rows, cols.by_path)
        return op.run_gathered(rows, cols.by_path)

    # preparation, as a prepared statement would pay it once: the first execution compiles the shape's kernels (hiprtc) and
    # agrees on the region capacities; then the untimed warm-up steps the driver asks for
    res, info = step()
    for _ in range(args.warmup):
        res, info = step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res, info = step()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    # the row all-to-all is the reported configuration; the partial-group exchange (G groups travel instead of
    # the rows) is measured next to it as the ablation, in the same run and with the same barriers
    ablation = None
    if info.get("mode") == "rows" and not op.has_distinct and not getattr(args, "no_ablation", False):
        try:
            for _ in range(args.warmup):
                ares, ainfo = op.run_gathered(rows, cols.by_path)
            torch.cuda.synchronize()
            dist.barrier()
            ta = time.perf_counter()
            for i in range(args.steps):
                ares, ainfo = op.run_gathered(rows, cols.by_path)
            torch.cuda.synchronize()
            dist.barrier()
            ael = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
            dist.all_reduce(ael, op=dist.ReduceOp.MAX)
            ael = float(ael.item())
            ablation = {"exchange": ainfo.get("mode"), "value": total_rows * args.steps / ael, "unit": "rows/s",
                        "ms_per_step": ael / args.steps * 1e3, "groups": int(ares["ngroups"]),
                        "what": "the same query with per-GPU partial groups exchanged instead of rows (SURVEY 8e's alternative for "
                                "low-cardinality keys); not the reported value"}
        except Exception as e:  # never costs the headline line
            ablation = {"error": repr(e)[:200]}
    # skew: what every owner received in the last step, and the per-destination capacities the ranks agreed on
    share = None
    if info.get("mode") == "rows":
        mine = torch.tensor([int(info.get("recv_rows", 0))], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        got = [int(x.item()) for x in every]
        mean = sum(got) / max(1, world)
        share = {"recv_rows": got, "largest_over_mean": (max(got) / mean) if mean else None, "region_rows": info.get("region_rows"),
                 "what": "rows every owner received in the last step; region capacities are per destination (a hot owner's regions alone grow)"}
    if rank == 0:
        how = {"gathered partials": "per-GPU partial groups merged after ONE RCCL all-gather (every rank holds the result)",
               "partials": "partial groups hash-partitioned on the group key by ONE RCCL all-to-all, final groups all-gathered",
               }.get(info.get("mode"), "filtered rows hash-partitioned on the group key by ONE RCCL all-to-all (counts + rows "
                                       "in one packed region per peer), final groups all-gathered")
        out = {
            "metric": bench.METRIC,
            "value": total_rows * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": bench.DTYPE,
            "data": "synthetic",
            "config": {"workload": "%s: %s @ %d rows in all over %d GPUs (%d per GPU), K_cat=%d, columns resident in HBM; %s" %
                                   (args.workload, wl["sql"], total_rows, world, rows, args.kcat, how),
                       "rows_per_gpu": rows, "total_rows": total_rows, "groups": int(res["ngroups"]), "exchange": info.get("mode"),
                       "exchange_rank0": info},
        }
        # the same definition as at N = 1, over the whole job: the query's algorithmic bytes on all ranks over the step's time
        # (max over ranks, barrier to barrier: the collectives are part of the query), against N x the per-GPU HBM peak
        alg = wl["bytes_per_row"] * total_rows
        step_ms = elapsed / args.steps * 1e3
        ach = alg / (step_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": bench.HBM_PEAK_GBS * world, "unit": "GB/s",
                           "frac": ach / (bench.HBM_PEAK_GBS * world), "traffic": None,
                           "what": "whole query over all ranks: algorithmic bytes / step time (max over ranks, exchange and gather included) "
                                   "against %d x %.0f GB/s" % (world, bench.HBM_PEAK_GBS),
                           "query_ms": step_ms, "algorithmic_bytes_per_launch": alg,
                           "kernel_split": {"rank 0, batch kernels": {
                               "ms": info.get("scan_ms"),
                               "what": "partition kernel (Filter + hash partition of the shard)" if info.get("mode", "").startswith("rows")
                               else "scan kernel of the shard"}}}
        if share is not None:
            out["owner_share"] = share
        if ablation is not None:
            out["ablation_partial_groups"] = ablation
        if not args.no_cpu:
            out["cpu_baseline"] = bench.cpu_baseline(wl, args.kcat, bool(args.zipf), total_rows, min(args.cpu_sample, total_rows))
        print(json.dumps(out))
    op.done()
    dist.destroy_process_group()
