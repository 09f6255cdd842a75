#!/bin/bash
# rocprofv3 passes over the headline bench (run on the GPU box from the repo root): kernel trace + stats, then the two
# PMC passes (each alone, as MI355X_MICROARCH.md prescribes).  Output under gpurun_out/prof/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/write.log 2>&1
ls $OUT/*
