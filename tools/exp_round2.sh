mkdir -p gpurun_out
B="python bench.py --no-cpu --no-ingest"
for i in 1 2; do
N1K_LIB=tools/ab/libn1k_base.so $B --three-calls > gpurun_out/z_c2_base_$i.log 2>&1
$B --three-calls > gpurun_out/z_c2_new3_$i.log 2>&1
$B > gpurun_out/z_c2_new1_$i.log 2>&1
done
$B --workload config3 > gpurun_out/z_c3_new.log 2>&1
$B --workload config5 > gpurun_out/z_c5_new.log 2>&1
$B --workload arith > gpurun_out/z_arith.log 2>&1
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t5.log 2>&1; echo rc=$? >> gpurun_out/t5.log
R="python bench.py --force-dist --exchange rows --no-cpu"
$R > gpurun_out/z_rows.log 2>&1
$R --opt part_block=256 > gpurun_out/z_rows_b256.log 2>&1
$R --opt part_block=256 --opt part_per_cu=4 > gpurun_out/z_rows_b256_4.log 2>&1
$R --opt part_block=256 --opt part_per_cu=6 > gpurun_out/z_rows_b256_6.log 2>&1
tail -3 gpurun_out/t5.log
for f in gpurun_out/z_*.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  ms_per_step %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
except Exception as e: print('  ERR', e)
"; done
