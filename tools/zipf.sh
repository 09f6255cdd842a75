#!/bin/bash
# config 2 at 100 M rows: uniform and Zipf(1.0) keys, K = 1000 and K = 16 (one line each: ms per step, scan kernel ms)
for a in "--kcat 1000 --zipf 0" "--kcat 1000 --zipf 1" "--kcat 16 --zipf 0" "--kcat 16 --zipf 1"; do
  python bench.py --no-ingest --no-sizes --no-cpu $a "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$a', 'ms_per_step %.4f query_ms %.4f batch_kernels_ms %.4f' % (d['ms_per_step'], r['query_ms'], r['kernel_split']['batch kernels']['ms']))"
done
