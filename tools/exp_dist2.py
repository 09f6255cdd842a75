"""Host-side stage timing of the stream-ordered partial-group step (world_size 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from query_amd import distributed as qd

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
wl = bench.workloads()["config2"]
cols = bench.DeviceColumns(rows, 1000, False, 0, rows, 0)
op = qd.ShardedFilterGroup(wl["cond"], wl["keys"], wl["aggs"], bench.synth_dictionary(1000), 0, 1, 0)
for _ in range(3):
    op.run_partials(rows, cols.by_path)
snd, rcv, lib = op.sender, op.receiver, op.sender._lib
send, recv = op._pbuf
cap = op.partial_capacity
acc = {}


def mark(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t


N = 50
torch.cuda.synchronize()
with torch.cuda.stream(op.stream):
    for _ in range(N):
        t = time.perf_counter()
        snd.reopen(); t = mark("snd.reopen", t)
        snd.process_device_items(rows, [cols.by_path[p] for p in op.send_paths]); t = mark("push", t)
        lib.n1k_export_partials_async(snd._h, 1, cap, send.data_ptr()); t = mark("export", t)
        dist.all_to_all_single(recv, send); t = mark("a2a", t)
        rcv.reopen(); t = mark("rcv.reopen", t)
        lib.n1k_merge_partials_device(rcv._h, 1, cap, recv.data_ptr()); t = mark("merge", t)
        raw = rcv.after_items_raw(); t = mark("finish(sync)", t)
        snd.sync(); t = mark("snd.sync", t)
        snd.stats(); t = mark("stats", t)
tot = 0
for k, v in acc.items():
    print("%-14s %8.1f us" % (k, v / N * 1e6))
    tot += v
print("total %.1f us" % (tot / N * 1e6))
dist.destroy_process_group()
