#!/bin/bash
# launch-parameter sweep of the headline workload (kernel_ms = scan + merge by HIP events)
for o in "grid_blocks=0" "grid_blocks=512" "grid_blocks=768" "grid_blocks=1024" "grid_blocks=1280" "grid_blocks=1536" "block=1024" "rows_per_lane=4" "slabs=0"; do
  python bench.py --steps 30 --warmup 3 --no-cpu --opt $o > gpurun_out/sw.json
  python - "$o" <<'PY'
import json,sys
d=json.load(open("gpurun_out/sw.json")); print(sys.argv[1], round(d["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4))
PY
done
