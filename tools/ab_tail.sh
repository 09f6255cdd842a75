# A/B of option tail_in_merge (merge + tail as one launch) at 100 M and 10 M rows
for i in 1 2 3; do for o in 0 1; do for r in 100000000 10000000; do
python bench.py --no-cpu --no-ingest --no-sizes --rows $r --opt tail_in_merge=$o 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('rows $r tail_in_merge=$o', round(d['ms_per_step'],4), round(d['roofline']['query_ms'],4))"
done; done; done
