#!/bin/bash
# One measurement round on the GPU box: kernel trace of the bench command, three PMC passes over its 1024-scan launches, then the bench line itself,
# summarised into gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/).
#   bash tools/profile_round.sh r03a             -> configs[1] (the driver's line)
#   bash tools/profile_round.sh r03a --config 3  -> files carry the suffix _config3
# Counters are collected in their own runs (--pmc with --kernel-trace only), as the pool requires.
set -u
TAG=$1; shift
SUF=""
case "$*" in *"--config 3"*) SUF=_config3;; *"--config 4"*) SUF=_config4;; *"--config 2"*) SUF=_config2;; esac
OUT=gpurun_out/prof_$TAG$SUF
mkdir -p $OUT
export TMPDIR=/tmp
export MML_LIB_SHA16=$(sha256sum multi-modal-loam_amd/libmmloam_hip.so | cut -c1-16)
ARGS="--steps 4 --warmup 1 --cpu-seconds 0 --strict $*"
# The counter passes run ONLY launches of 1024 scans on one stream (the launches bench.py times for its roofline object): the
# summarisers average a kernel's launches of the largest grid, and the kernels with a fixed grid (k_associate, the k_select rows) look
# the same at every batch size -- mixed with the timed region's per-lane launches their counters would describe no launch at all.
PMCARGS="--steps 1 --warmup 0 --cpu-seconds 0 --slots ${KB:-1024} --batch ${KB:-1024} --skip-upload --skip-by-slots $*"
export PMC_LANES=1
rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
DB=$(ls $OUT/trace/*/*.db 2>/dev/null | head -1); [ -z "$DB" ] && DB=$(ls $OUT/trace/*.db 2>/dev/null | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/${TAG}_kernel_stats$SUF.md
if [ "$SUF" != "_config2" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- env MML_LANES=$PMC_LANES python bench.py $PMCARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- env MML_LANES=$PMC_LANES python bench.py $PMCARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -o s -- env MML_LANES=$PMC_LANES python bench.py $PMCARGS > /dev/null 2> $OUT/pmc_sq.err
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
S=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1)
KB=${KB:-1024}   # scans per launch of the counter passes (= --slots / --batch of PMCARGS)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_traffic.py $F $W $KB > $OUT/traffic_${TAG%%[a-z]}$SUF.json
[ -n "$S" ] && python tools/pmc_sq.py $S --json $OUT/sq_${TAG%%[a-z]}$SUF.json $KB > $OUT/${TAG}_sq_counters$SUF.md
fi
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
# the bench line last, against the counter files just taken.  SIDE EFFECT, on purpose: they are copied into profiles/ of THIS
# checkout (on the GPU box: the scratch copy) so that the line carries their traffic and instruction counts as current, not
# stale; what is to be committed still has to be copied from gpurun_out/prof_<tag>/ into profiles/ by hand.
for f in $OUT/traffic_*.json $OUT/sq_*.json; do [ -s "$f" ] && cp $f profiles/; done
python bench.py --strict $* > $OUT/${TAG}_bench$SUF.json 2> $OUT/bench.err
echo "bench rc=$?" >> $OUT/bench.err
ls -la $OUT; tail -c 400 $OUT/${TAG}_bench$SUF.json; echo; head -30 $OUT/${TAG}_kernel_stats$SUF.md | cut -c1-150
