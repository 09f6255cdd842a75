# A/B/C on one box: the committed baseline (tools/ab/libn1k_base.so), the current build, and tools/ab/libn1k_pipe.so
mkdir -p gpurun_out
for i in 1 2 3; do
 for lib in tools/ab/libn1k_base.so query_amd/libn1k.so tools/ab/libn1k_pipe.so; do
  N1K_LIB=$lib python bench.py --no-cpu --no-ingest --no-sizes "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', 'ms', round(d['ms_per_step'],4), 'query', round(r['query_ms'],4), 'batch', round(r['kernel_split']['batch kernels']['ms'],4))"
 done
done
