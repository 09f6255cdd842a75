"""Multi-GPU Filter -> Group -> Aggregate: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed query path (README.md:61-65); inside one process it fans Parallel copies into
one serial IntermediateGroup through an in-memory queue (execution/exchange.go:161-251).  Here every rank

  1. filters its own row shard and hash-partitions the survivors on the group key
     (n1k_partition_device_batch: hash(key) % world, one region per destination rank),
  2. exchanges the regions with ONE all-to-all per column buffer (RCCL all_to_all_single with split sizes; on an
     8-GPU node every GPU pair has its own xGMI link, so all 7 links of a GPU carry traffic at once),
  3. runs InitialGroup/FinalGroup on the rows it received (it owns those groups entirely, so no partial states
     cross the fabric and COUNT(DISTINCT) needs no set merge),
  4. gathers the finished groups on rank 0 (the only serial step, G rows).

The exchange code is backend-agnostic (it moves torch tensors with all_to_all_single / all_gather), so the same
functions run under gloo on CPU tensors in tests/test_distributed_cpu.py.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import time
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def exchange_counts(counts, group=None):
    """counts[d] rows this rank sends to rank d  ->  tensor recv[s] rows it receives from rank s."""
    import torch
    import torch.distributed as dist
    recv = torch.empty_like(counts)
    dist.all_to_all_single(recv, counts, group=group)
    return recv


def exchange_rows(send_cols: Sequence, send_counts: Sequence[int], recv_counts: Sequence[int], group=None) -> List:
    """One all_to_all_single per column buffer.  send_cols[c] is contiguous and ordered by destination rank
    (send_counts rows each); returns the received column buffers ordered by source rank."""
    import torch
    import torch.distributed as dist
    out = []
    total = int(sum(recv_counts))
    for col in send_cols:
        recv = torch.empty((total,) + tuple(col.shape[1:]), dtype=col.dtype, device=col.device)
        dist.all_to_all_single(recv, col, output_split_sizes=[int(x) for x in recv_counts],
                               input_split_sizes=[int(x) for x in send_counts], group=group)
        out.append(recv)
    return out


def gather_groups(local: np.ndarray, device=None, dst: int = 0, group=None) -> Optional[np.ndarray]:
    """Gather variable-length arrays of fixed-size records (uint8 [n, record_bytes]) on rank dst."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    width = local.shape[1] if local.ndim == 2 else 1
    mx = max(max(counts), 1)
    pad = np.zeros((mx, width), dtype=np.uint8)
    pad[:local.shape[0]] = local.reshape(local.shape[0], width)
    mine = torch.from_numpy(pad).to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)  # G rows: tiny next to the row exchange
    if rank != dst:
        return None
    parts = [bufs[r][:counts[r]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, width), np.uint8)


class FixedGather:
    """Gather of the finished groups with ONE collective: every rank contributes a fixed-size slot
    [count i64][records ...]; if some rank has more records than a slot holds, every rank sees it in the gathered
    headers and all retry with larger slots."""

    def __init__(self, record_bytes: int, capacity: Optional[int] = None):
        """capacity: records per slot, THE SAME ON EVERY RANK (a slot's size is part of the collective's shape); None =
        agreed on at the first call from the largest contribution (one small all-reduce)."""
        self.record_bytes = max(int(record_bytes), 1)
        self.capacity = capacity
        self._bufs = None

    def __call__(self, local: np.ndarray, device, group=None) -> np.ndarray:
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(group)
        n = int(local.shape[0])
        if self.capacity is None:
            # ranks own different numbers of groups after a hash partition: the slot size must not depend on the local one
            most = torch.tensor([n], dtype=torch.int64, device=device)
            dist.all_reduce(most, op=dist.ReduceOp.MAX, group=group)
            self.capacity = max(1024, int(2 ** int(np.ceil(np.log2(max(2 * int(most.item()), 1))))))
        while True:
            slot = 8 + self.capacity * self.record_bytes
            if self._bufs is None or self._bufs[0].numel() != slot:
                self._bufs = (torch.zeros(slot, dtype=torch.uint8, device=device),
                              torch.zeros(slot * world, dtype=torch.uint8, device=device))
            mine, allb = self._bufs
            host = np.zeros(slot, dtype=np.uint8)
            host[:8] = np.array([n], dtype=np.int64).view(np.uint8)
            k = min(n, self.capacity)
            host[8:8 + k * self.record_bytes] = local[:k].reshape(-1)
            mine.copy_(torch.from_numpy(host))
            dist.all_gather_into_tensor(allb, mine, group=group)
            got = allb.cpu().numpy().reshape(world, slot)
            counts = got[:, :8].copy().view(np.int64).reshape(world)
            if counts.max() <= self.capacity:
                parts = [got[r, 8:8 + int(counts[r]) * self.record_bytes].reshape(int(counts[r]), self.record_bytes)
                         for r in range(world)]
                return np.concatenate(parts, axis=0)
            self.capacity = int(2 ** int(np.ceil(np.log2(counts.max()))))


class ShardedFilterGroup:
    """One rank's share of the distributed operator (device path through libn1k.so)."""

    def __init__(self, condition: Optional[str], keys: Sequence[str], aggs: Sequence[str], dictionary: Sequence[bytes],
                 rank: int, world: int, device: int, order=None, limit=None, offset=None, having=None, **options):
        import query_amd
        from query_amd import plan
        self.rank, self.world, self.device = rank, world, device
        self.tail = order is not None or limit is not None or offset is not None or having is not None
        # sender: Filter + key evaluation; receiver: the same grouping without the Filter (it was applied already)
        import torch
        # one stream for both handles and for torch (RCCL work is ordered against it by c10d's events)
        self.stream = torch.cuda.Stream(device=torch.device("cuda", device))
        self.sender = query_amd.GpuFilterGroup(plan.filter_group_plan(condition, keys, aggs), device=device,
                                               stream=self.stream.cuda_stream)
        # The grouped tail (HAVING, ORDER BY, OFFSET, LIMIT) belongs to whoever holds COMPLETE groups: `receiver` is the
        # owner of a hash range (row / partial-group exchange) and keeps its first offset+limit rows after HAVING;
        # `merger` holds the exact tail: it merges all ranks' partial groups in the gathered mode, and orders the
        # union of the owners' rows otherwise (n1k_order_rows).
        owner_limit = None if limit is None else int(limit) + int(offset or 0)
        self.receiver = query_amd.GpuFilterGroup(
            plan.filter_group_plan(None, keys, aggs, order=order, limit=owner_limit, having=having) if self.tail
            else plan.filter_group_plan(None, keys, aggs), device=device, stream=self.stream.cuda_stream, **options)
        self.merger = query_amd.GpuFilterGroup(
            plan.filter_group_plan(None, keys, aggs, order=order, limit=limit, offset=offset, having=having),
            device=device, stream=self.stream.cuda_stream, **options) if self.tail else self.receiver
        self._recv_ready = set()
        self._sbatch = None
        self._cap_known = False
        self._gather = None
        self.sender.intern(list(dictionary))
        self.receiver.intern(list(dictionary))
        if self.merger is not self.receiver:
            self.merger.intern(list(dictionary))
        self.send_paths = self.sender.column_paths
        self.recv_paths = self.receiver.column_paths
        self._bufs = None
        self.has_distinct = any("(distinct " in a.lower() for a in aggs)
        self.partial_capacity = int(options.get("partial_capacity", 4096)) if options else 4096
        self._pbuf = None

    # ------------------------------------------------------------------ exchange of partial groups
    # partial groups of ALL ranks fit this many bytes per rank -> gather them instead of partitioning them
    GATHER_LIMIT = 32 << 20

    def run_gathered(self, nrows: int, cols_by_path: Dict[str, tuple], want_rows_selected: bool = True) -> Tuple[dict, dict]:
        """Few groups: every rank aggregates its shard, ONE all_gather moves the per-GPU partial groups and every
        rank merges all of them (≙ IntermediateGroup + FinalGroup, replicated): each rank ends with the complete
        result, so no second collective is needed.  Falls over to the hash-partitioned exchange (run_partials) when
        world x region outgrows GATHER_LIMIT."""
        return self.run_partials(nrows, cols_by_path, want_rows_selected, replicate=True)

    def run_partials(self, nrows: int, cols_by_path: Dict[str, tuple], want_rows_selected: bool = True,
                     replicate: bool = False) -> Tuple[dict, dict]:
        """Few groups next to rows: aggregate the local shard first (same kernels as on one GPU), then move only
        the partial groups: ONE all_to_all_single of fixed-size regions, hash-partitioned on the group key, merged
        by the owner (≙ IntermediateGroup) and finalised there.

        The whole step is stream-ordered — scan, export, all-to-all, merge, finalize — with a single host
        synchronisation at the end (n1k_finish's copy of the groups).  A sender that cannot export (a region
        overflowed; keys coded by device-local value tables) says so in every region header, the merge on each
        receiver then does nothing and n1k_finish reports it: all ranks take the same retry branch without an
        extra collective."""
        import torch
        import torch.distributed as dist
        from query_amd import _ffi
        from query_amd.gpu_operator import N1kError
        snd = self.sender  # Filter + InitialGroup over the shard
        lib = snd._lib
        dev = torch.device("cuda", self.device)
        with torch.cuda.stream(self.stream):
            while True:
                cap = self.partial_capacity
                region = int(lib.n1k_partial_region_bytes(snd._h, cap))
                # the same on every rank: capacities only change on verdicts that all ranks see
                gathered = replicate and region * self.world <= self.GATHER_LIMIT
                rcv = self.merger if gathered else self.receiver  # merge + FinalGroup (+ the grouped tail)
                if id(rcv) not in self._recv_ready:  # the merging handle needs the key layout (column kinds) first
                    rcv.process_device_items(0, [cols_by_path[p] for p in self.recv_paths])
                    self._recv_ready.add(id(rcv))
                nsend = 1 if gathered else self.world
                if self._pbuf is None or self._pbuf[0].numel() != region * nsend or self._pbuf[1].numel() != region * self.world:
                    self._pbuf = (torch.empty(region * nsend, dtype=torch.uint8, device=dev),
                                  torch.empty(region * self.world, dtype=torch.uint8, device=dev))
                send, recv = self._pbuf
                snd.reopen()
                if self._sbatch is None or self._sbatch[0] is not cols_by_path or self._sbatch[1] != nrows:
                    # the n1k_batch of a shard that stays where it is (same dict object, same rows) is built once
                    self._sbatch = (cols_by_path, nrows, snd.make_device_batch(nrows, [cols_by_path[p] for p in self.send_paths]))
                snd.process_device_batch(self._sbatch[2])
                if not self._cap_known:
                    # first step: size the regions from the number of groups the shards really hold (one small
                    # all-reduce) instead of climbing there by x4 retries that each repeat the scan
                    snd.sync()
                    ng = torch.tensor([int(snd.stats()["groups_out"])], dtype=torch.int64, device=dev)
                    dist.all_reduce(ng, op=dist.ReduceOp.MAX)
                    ng = int(ng.item())
                    self._cap_known = True
                    full = 1 << max(12, int(np.ceil(np.log2(ng * 1.25 + 64))))
                    if replicate and int(lib.n1k_partial_region_bytes(snd._h, full)) * self.world <= self.GATHER_LIMIT:
                        want = full
                    else:
                        want = 1 << max(12, int(np.ceil(np.log2(ng / self.world * 1.5 + 64))))
                    if want != cap:
                        self.partial_capacity = want
                        continue
                snd._check(lib.n1k_export_partials_async(snd._h, nsend, cap, send.data_ptr()))
                if gathered:
                    dist.all_gather_into_tensor(recv, send)  # region r = rank r's partial groups, on every rank
                else:
                    dist.all_to_all_single(recv, send)  # equal splits: region d goes to rank d
                rcv.reopen()
                rcv._check(lib.n1k_merge_partials_device(rcv._h, self.world, cap, recv.data_ptr()))
                try:
                    raw = rcv.after_items_raw()  # the step's one host synchronisation
                except N1kError as e:
                    if e.status == _ffi.OOM and "region" in e.message:
                        self.partial_capacity *= 4
                        continue
                    if e.status == _ffi.UNSUPPORTED:
                        snd.reopen()  # drop the abandoned export's groups and flags
                        raw, info = self.run(nrows, cols_by_path)
                        info["mode"] = "rows (wide key values)"
                        return raw, info
                    raise
                break
            if want_rows_selected:
                snd.sync()  # one more small copy: the Filter's survivor count lives in the sender's counters
            stats = snd.stats()  # scan time from the completed HIP events (no waiting)
        return raw, {"mode": "gathered partials" if gathered else "partials", "region_bytes": region,
                     "rows_selected": int(stats["rows_selected"]) if want_rows_selected else None,
                     "scan_ms": float(stats["device_ms"]), "spec_kernel": int(stats["spec_kernel"])}

    def combine(self, raw: dict, info: dict, device) -> Tuple[np.ndarray, np.ndarray]:
        """The job's result on every rank as (keys, aggs) structured arrays: in the gathered mode the merging handle
        already holds it; otherwise the owners' rows are gathered (one all-gather) and, when the plan has a grouped
        tail, ordered and cut by the merger handle (n1k_order_rows)."""
        from query_amd.gpu_operator import GpuFilterGroup
        if info.get("mode") == "gathered partials":
            return raw["keys"], raw["aggs"]
        nk, na, ng = raw["nkeys"], raw["naggs"], raw["ngroups"]
        rec = np.concatenate([raw["keys"].view(np.uint8).reshape(ng, -1), raw["aggs"].view(np.uint8).reshape(ng, -1)], axis=1) \
            if ng else np.zeros((0, 16 * (nk + na)), np.uint8)
        if self._gather is None:
            self._gather = FixedGather(16 * (nk + na))  # capacity agreed on across the ranks at the first call
        allg = np.ascontiguousarray(self._gather(rec, device))
        n = allg.shape[0]
        dt = GpuFilterGroup._VALUE_DT
        keys = np.ascontiguousarray(allg[:, :16 * nk]).view(dt).reshape(n, nk)
        aggs = np.ascontiguousarray(allg[:, 16 * nk:]).view(dt).reshape(n, na)
        if self.tail:
            r = self.merger.order_rows(keys, aggs)
            return r["keys"], r["aggs"]
        return keys, aggs

    def _alloc(self, capacity: int, kinds: Sequence[int]):
        import torch
        from query_amd import _ffi
        dev = torch.device("cuda", self.device)
        bufs = []
        for k in kinds:
            if k == _ffi.COL_DICT32:
                bufs.append({"codes": torch.empty(capacity * self.world, dtype=torch.int32, device=dev)})
            else:
                bufs.append({"tags": torch.empty(capacity * self.world, dtype=torch.uint8, device=dev),
                             "payload": torch.empty(capacity * self.world, dtype=torch.int64, device=dev)})
        counts = torch.zeros(self.world, dtype=torch.int64, device=dev)
        return bufs, counts

    def run(self, nrows: int, cols_by_path: Dict[str, tuple]) -> Tuple[dict, dict]:
        """cols_by_path: path -> (kind, tags_ptr, payload_ptr, codes_ptr) device addresses of this rank's shard.
        Returns (local final groups as numpy record arrays, timing/volume info)."""
        import torch
        with torch.cuda.stream(self.stream):
            return self._run_rows(nrows, cols_by_path)

    def _run_rows(self, nrows: int, cols_by_path: Dict[str, tuple]) -> Tuple[dict, dict]:
        import torch
        import torch.distributed as dist
        from query_amd import _ffi
        cols = [cols_by_path[p] for p in self.send_paths]
        kinds = [c[0] for c in cols]
        if self._bufs is None or self._bufs[2] < nrows:
            self._bufs = self._alloc(nrows, kinds) + (nrows,)
        bufs, counts, cap = self._bufs
        # 1. filter + hash partition on the group key
        batch, keep = self.sender._make_batch(nrows, cols)
        out_arr = (_ffi.Col * len(cols))()
        for i, (k, b) in enumerate(zip(kinds, bufs)):
            out_arr[i].kind = k
            if k == _ffi.COL_DICT32:
                out_arr[i].codes = b["codes"].data_ptr()
            else:
                out_arr[i].tags = b["tags"].data_ptr()
                out_arr[i].payload = b["payload"].data_ptr()
        torch.cuda.synchronize()
        st = self.sender._lib.n1k_partition_device_batch(self.sender._h, C.byref(batch), self.world, cap, out_arr,
                                                        counts.data_ptr())
        self.sender._check(st)
        # 2. all-to-all over xGMI: counts, then one collective per column buffer
        recv_counts = exchange_counts(counts)
        sc = counts.cpu().tolist()
        rc = recv_counts.cpu().tolist()
        send_cols, layout = [], []
        for k, b in zip(kinds, bufs):
            for name in (("codes",) if k == _ffi.COL_DICT32 else ("tags", "payload")):
                t = b[name]
                send_cols.append(torch.cat([t[d * cap: d * cap + sc[d]] for d in range(self.world)]))
                layout.append(name)
        recv = exchange_rows(send_cols, sc, rc)
        # 3. local InitialGroup .. FinalGroup on the rows this rank owns
        received = {}
        it = iter(recv)
        for p, k in zip(self.send_paths, kinds):
            if k == _ffi.COL_DICT32:
                t = next(it)
                received[p] = (k, None, None, t.data_ptr(), t)
            else:
                tg, pl = next(it), next(it)
                received[p] = (k, tg.data_ptr(), pl.data_ptr(), None, (tg, pl))
        nrecv = int(sum(rc))
        self.receiver.reopen()
        torch.cuda.synchronize()
        self.receiver.process_device_items(nrecv, [received[p][:4] for p in self.recv_paths])
        raw = self.receiver.after_items_raw()
        info = {"sent_rows": int(sum(sc)), "recv_rows": nrecv}
        return raw, info


def bench_main(args, rank: int, world: int, local_rank: int):
    """bench.py --gpus N (N > 1): weak scaling, every rank owns `--rows` rows of the global data set."""
    import torch
    import torch.distributed as dist
    import bench
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "MASTER_ADDR" not in os.environ:  # --force-dist without a launcher
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    wl = bench.workloads()[args.workload]
    total_rows = args.rows * world
    cols = bench.DeviceColumns(args.rows, args.kcat, bool(args.zipf), rank * args.rows, total_rows, local_rank)
    op = ShardedFilterGroup(wl["cond"], wl["keys"], wl["aggs"], bench.synth_dictionary(args.kcat), rank, world,
                            local_rank, order=wl.get("order"), limit=wl.get("limit"))
    dev = torch.device("cuda", local_rank)

    def step(last=False):
        if op.has_distinct or args.exchange == "rows":
            raw, info = op.run(args.rows, cols.by_path)
        elif args.exchange == "partials":
            raw, info = op.run_partials(args.rows, cols.by_path, want_rows_selected=last)
        else:
            raw, info = op.run_gathered(args.rows, cols.by_path, want_rows_selected=last)
        keys, _aggs = op.combine(raw, info, dev)  # every rank holds the job's result; rank 0 reports it
        return keys, info

    for _ in range(args.warmup):
        allg, info = step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        allg, info = step(last=(i == args.steps - 1))
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    if rank == 0:
        mode = info.get("mode", "rows")
        how = ("per-GPU partial groups merged after ONE RCCL all-gather (every rank holds the result)" if mode == "gathered partials"
               else "partial groups hash-partitioned on the group key by RCCL all-to-all, final groups gathered on rank 0"
               if mode == "partials" else
               "filtered rows hash-partitioned on the group key by RCCL all-to-all, final groups gathered on rank 0")
        out = {
            "metric": "rows/sec filter+group-by on 100M synthetic JSON docs; achieved HBM GB/s",
            "value": total_rows * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64/f64 tagged scalars (u8 tag + 8 B payload), u32 dictionary codes",
            "data": "synthetic",
            "config": {"workload": "%s: %s @ %d rows per GPU x %d GPUs, K_cat=%d, %s" %
                                   (args.workload, wl["sql"], args.rows, world, args.kcat, how),
                       "rows_per_gpu": args.rows, "groups": int(allg.shape[0]) if allg is not None else None,
                       "exchange_rank0": info},
        }
        if info.get("scan_ms"):
            alg = wl["bytes_per_row"] * args.rows
            ach = alg / (info["scan_ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": bench.HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / bench.HBM_PEAK_GBS, "traffic": None, "kernel_ms": info["scan_ms"],
                               "kernel": "rank 0: scan_spec_kernel(+merge_slabs_kernel)" if info.get("spec_kernel")
                               else "rank 0: scan kernel", "algorithmic_bytes_per_launch": alg}
        print(json.dumps(out))
    dist.destroy_process_group()
