"""Stage timing of one multi-GPU step at world_size 1 (same code path as N > 1): where the latency goes."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import bench
from query_amd import distributed as qd

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "partials"
wl = bench.workloads()["config2"]
cols = bench.DeviceColumns(rows, 1000, False, 0, rows, 0)
op = qd.ShardedFilterGroup(wl["cond"], wl["keys"], wl["aggs"], bench.synth_dictionary(1000), 0, 1, 0)
marks = []
_orig = time.perf_counter


def run():
    return op.run_partials(rows, cols.by_path) if mode == "partials" else op.run(rows, cols.by_path)


for _ in range(3):
    run()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    run()
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
