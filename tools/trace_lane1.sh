#!/bin/bash
# kernel trace of the bench's timed region on ONE stream (every launch covers all resident slots): gpurun_out/<tag>/kernel_stats_lane1.md
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- env MML_LANES=1 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-upload --kernel-steps 1 $* > $OUT/bench_lane1.json 2> $OUT/trace.err
DB=$(ls $OUT/trace/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats_lane1.md
rm -rf $OUT/trace
head -45 $OUT/kernel_stats_lane1.md | cut -c1-160
