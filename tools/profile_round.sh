#!/bin/bash
# tools/profile_round.sh <round tag, e.g. r02>: the round's committed evidence for every bench workload, run on the GPU box
# from the repo root — per workload the bench line, the rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE,
# WRITE_SIZE, each alone).  Output: gpurun_out/profiles_<tag>/ — copy into profiles/ and commit.
set -e
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $DST
# (the first kernel of a query, as the timeline's marker: one-call executions of small-table plans start with their scan — the
#  last kernel of the query before left the device as a reopen would; config 3 / 5 still start with the reopen kernel)
declare -A FIRST=( [config2]=scan_spec_kernel [config3]=init_table_kernel [config5]=init_table_kernel [arith]=n1k_jit_wide [filter]=init_table_kernel )
for WL in config2 config3 config5 arith filter; do
  python3 $ROOT/bench.py --workload $WL --steps 20 --warmup 3 $([ $WL = config2 ] || echo --no-sizes) 2>/dev/null | tail -1 > $DST/${TAG}_bench_${WL}_100M.json.log
  bash $ROOT/tools/prof.sh ${TAG}_$WL all -- --workload $WL > /dev/null
  cp $ROOT/gpurun_out/prof_${TAG}_$WL/stats/bench_kernel_stats.csv $DST/${TAG}_bench_${WL}_100M_kernel_stats.csv
  cp $ROOT/gpurun_out/prof_${TAG}_$WL/pmc_fetch_write.json $DST/${TAG}_bench_${WL}_100M_pmc_fetch_write.json
  python3 $ROOT/tools/timeline.py $ROOT/gpurun_out/prof_${TAG}_$WL/stats ${FIRST[$WL]} 1 > $DST/${TAG}_bench_${WL}_100M_timeline.txt
  echo "$WL done"
done
# the multi-GPU row exchange with one rank (the same code path: partition -> all-to-all -> owner's scan -> gather)
python3 $ROOT/bench.py --force-dist --exchange rows --steps 20 --warmup 3 2>/dev/null | tail -1 > $DST/${TAG}_bench_rows_world1_100M.json.log || true
bash $ROOT/tools/prof.sh ${TAG}_rows stats -- --force-dist --exchange rows > /dev/null || true
cp $ROOT/gpurun_out/prof_${TAG}_rows/stats/bench_kernel_stats.csv $DST/${TAG}_bench_rows_world1_100M_kernel_stats.csv || true
ls -la $DST
