#!/bin/bash
# tools/sweep_opts.sh <workload> "<opt=val[,opt=val]>" ... : ms per step of bench.py under engine option sets
WL=$1; shift
for o in "$@"; do
  args=""
  IFS=',' read -ra parts <<< "$o"
  for p in "${parts[@]}"; do [ -n "$p" ] && [ "$p" != "-" ] && args="$args --opt $p"; done
  python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu $args > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "$o FAILED"; tail -3 gpurun_out/sw.err; continue; }
  python - "$o" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/sw.json").read().strip().splitlines()[-1]); print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "scan_ms", round(d["roofline"]["kernel_ms"],4))
PY
done
