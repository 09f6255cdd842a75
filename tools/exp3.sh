# tools/exp3.sh <workload> "<opts for run 1>" "<opts for run 2>" ...   (each run: rocprofv3 kernel stats of bench.py)
WLD=$1; shift
mkdir -p gpurun_out
for o in "$@"; do
  args=""; for x in $o; do args="$args --opt $x"; done
  bash tools/prof.sh e3 stats -- --workload $WLD $args > /dev/null 2>&1
  echo "== $o"; grep -E "agg_bins16|radix_scatter_sub|scan_spec_records|scan_spec_kernel|dedupe|filter_stream" gpurun_out/prof_e3/kernel_stats.txt | awk '{print "   ", $(NF-2), substr($0,1,60)}'
  grep '^{' gpurun_out/prof_e3/stats.log | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('    ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('ERR', e)"
done
