#!/bin/bash
# rocprofv3 passes over one bench.py workload (run on the GPU box from the repo root):
#   tools/prof.sh <tag> [stats|pmc|all] -- <bench.py arguments>
# kernel trace + stats, and the two PMC passes (each alone, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/prof_<tag>/.
set -e
TAG=$1; MODE=$2; shift 2; [ "$1" = "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = "stats" ] || [ "$MODE" = "all" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ingest --no-sizes "$@" > $OUT/stats.log 2>&1
  python3 $ROOT/tools/summarise_stats.py $OUT/stats > $OUT/kernel_stats.txt
fi
if [ "$MODE" = "pmc" ] || [ "$MODE" = "all" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-ingest --no-sizes "$@" > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-ingest --no-sizes "$@" > $OUT/write.log 2>&1
  python3 $ROOT/tools/summarise_pmc.py $OUT 4 > $OUT/pmc_fetch_write.json
fi
if [ "$MODE" = "sq" ]; then
  # where the waves' cycles go (one pass: 8 SQ slots) — per kernel: wave-cycles parked / issue-stalled / active, instruction mix
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-ingest --no-sizes "$@" > $OUT/sq.log 2>&1
  python3 $ROOT/tools/summarise_sq.py $OUT/sq > $OUT/sq.txt
fi
ls $OUT
