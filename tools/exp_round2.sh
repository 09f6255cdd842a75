mkdir -p gpurun_out
B="python bench.py --no-cpu --no-ingest"
$B --rows 10000000 > gpurun_out/v_c2_10M.log 2>&1
$B --rows 1000000000 --steps 10 > gpurun_out/v_c2_1B.log 2>&1
$B --workload config3 --rows 10000000 > gpurun_out/v_c3_10M.log 2>&1
$B --workload config5 --rows 10000000 > gpurun_out/v_c5_10M.log 2>&1
R="python bench.py --force-dist --exchange rows --no-cpu --no-ablation"
$R > gpurun_out/v_rows.log 2>&1
N1K_JIT_PART_NOPIPE=1 $R --opt part_per_cu=8 > gpurun_out/v_rows_nopipe8.log 2>&1
N1K_JIT_PART_NOPIPE=1 $R --opt part_per_cu=6 > gpurun_out/v_rows_nopipe6.log 2>&1
N1K_JIT_PART_NOPIPE=1 $R --opt part_per_cu=5 > gpurun_out/v_rows_nopipe5.log 2>&1
python bench.py --force-dist --exchange rows --no-cpu > gpurun_out/v_rows_abl.log 2>&1
for f in gpurun_out/v_*.log; do echo $f; tail -1 $f | python3 -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('  ms_per_step %.4f value %.4g kernel_ms %.4f frac %.3f %s' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('ablation_partial_groups')))
except Exception as e: print('  ERR', e)
"; done
