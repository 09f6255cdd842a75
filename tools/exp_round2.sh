mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_plans.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/t3.log 2>&1; echo rc=$? >> gpurun_out/t3.log
B="python bench.py --no-cpu --no-ingest"
for w in arith arith_filter arith_sum arith_plain; do $B --workload $w > gpurun_out/x_$w.log 2>&1; done
$B > gpurun_out/x_c2.log 2>&1
$B --opt flag_bytes=0 > gpurun_out/x_c2_noflagbytes.log 2>&1
$B --opt pinned_out=0 > gpurun_out/x_c2_nopinned.log 2>&1
$B --workload config2_allaggs > gpurun_out/x_c2all.log 2>&1
$B --workload config2_allaggs --opt flag_bytes=0 > gpurun_out/x_c2all_noflagbytes.log 2>&1
R="python bench.py --force-dist --exchange rows --no-cpu"
$R > gpurun_out/x_rows2.log 2>&1
$R --opt part_per_cu=3 > gpurun_out/x_rows3.log 2>&1
N1K_JIT_PART_ATTR='__attribute__((amdgpu_waves_per_eu(6,6)))' $R --opt part_per_cu=3 > gpurun_out/x_rows3w6.log 2>&1
N1K_JIT_PART_ATTR='__attribute__((amdgpu_waves_per_eu(8,8)))' $R --opt part_per_cu=4 > gpurun_out/x_rows4w8.log 2>&1
$R --opt jit=0 > gpurun_out/x_rows_interp.log 2>&1
tail -3 gpurun_out/t3.log
for f in gpurun_out/x_*.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception as e: print('  ERR', e)
"; done
