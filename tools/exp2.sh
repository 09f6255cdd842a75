set -o pipefail
mkdir -p gpurun_out
for o in "$@"; do
  bash tools/prof.sh e2 stats -- --workload ${WL:-config5} --opt $o > /dev/null 2>&1
  echo "== $o"; grep -E "agg_bins16|radix_scatter_sub|scan_spec_records|scan_spec_kernel|dedupe" gpurun_out/prof_e2/kernel_stats.txt | awk '{print "   ", $(NF-2), substr($0,1,60)}'
  grep '^{' gpurun_out/prof_e2/stats.log | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('    ms_per_step', round(d['ms_per_step'],4))
except Exception as e: print('ERR', e)"
done
