#!/bin/bash
# One measurement round on the GPU box: kernel trace + the three PMC passes of the default bench command, summarised
# into profiles/<tag>_*.  Usage (inside gpurun): bash tools/profile_round.sh r02a [bench args...]
# Counters are collected in their own runs (--pmc with --kernel-trace only), as the pool requires.
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --cpu-seconds 0 $*"
python bench.py --steps 20 --warmup 5 $* > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
DB=$(ls $OUT/trace/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats.md
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o s -- python bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
S=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W 1024 > $OUT/traffic_$TAG.json
[ -n "$S" ] && python tools/pmc_sq.py $S > $OUT/${TAG}_sq_counters.md
ls -la $OUT; tail -c 600 $OUT/bench.json; echo; head -40 $OUT/${TAG}_kernel_stats.md; cat $OUT/traffic_$TAG.json; cat $OUT/${TAG}_sq_counters.md
