# A/B on one box: tools/ab/libn1k_base.so against the current build, one bench workload per argument string
mkdir -p gpurun_out
for w in "$@"; do
 for i in 1 2; do
  for lib in tools/ab/libn1k_base.so query_amd/libn1k.so; do
   N1K_LIB=$lib python bench.py --no-cpu --no-ingest --no-sizes $w 2>/dev/null | tail -1 | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '$lib'.split('/')[-1], 'ms', round(d['ms_per_step'],4), 'query', round(r.get('query_ms',0),4))
except Exception as e: print('$w $lib ERR', e)"
  done
 done
done
