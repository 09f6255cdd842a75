#!/bin/bash
# tools/ab.sh — A/B of two builds of the library on ONE GPU box (boxes of the pool differ by ~ +-8 %, so numbers from
# different gpurun calls do not compare).  Build the baseline into tools/ab/ first (git-ignored, travels with the snapshot):
#   git worktree add /tmp/wt <commit> && (cd /tmp/wt && python -m query_amd.build) && cp /tmp/wt/query_amd/libn1k.so tools/ab/libn1k_base.so
# then on the box:  bash tools/ab.sh [bench.py arguments]
# Older builds lack n1k_run_device_batch: both sides are driven through the three calls (--three-calls).
mkdir -p gpurun_out
B="python bench.py --no-cpu --no-ingest --three-calls $@"
for i in 1 2 3; do
  N1K_LIB=tools/ab/libn1k_base.so $B > gpurun_out/ab_base_$i.log 2>&1
  $B > gpurun_out/ab_new_$i.log 2>&1
done
for f in gpurun_out/ab_*.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  ms_per_step %.4f query_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline'].get('query_ms', d['roofline'].get('kernel_ms', 0)), d['roofline']['frac']))
except Exception as e: print('  ERR', e)
"; done
