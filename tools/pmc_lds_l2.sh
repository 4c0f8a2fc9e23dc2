#!/bin/bash
# LDS bank-conflict and L2 hit-rate counters of the 1024-scan launches of one bench configuration (two --pmc passes with
# --kernel-trace only, as the pool requires):  bash tools/pmc_lds_l2.sh r05f [--config 3]  -> gpurun_out/prof_<tag>*/<tag>_lds_l2_counters*.md
set -u
TAG=$1; shift
SUF=""
case "$*" in *"--config 3"*) SUF=_config3;; *"--config 4"*) SUF=_config4;; esac
OUT=gpurun_out/prof_$TAG$SUF; mkdir -p $OUT
export TMPDIR=/tmp
PMCARGS="--steps 1 --warmup 0 --cpu-seconds 0 --slots ${KB:-1024} --batch ${KB:-1024} --skip-upload $*"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_lds -o l -- env MML_LANES=1 python bench.py $PMCARGS > /dev/null 2> $OUT/pmc_lds.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o t -- env MML_LANES=1 python bench.py $PMCARGS > /dev/null 2> $OUT/pmc_tcc.err
L=$(find $OUT/pmc_lds -name "*counter_collection.csv" | head -1); T=$(find $OUT/pmc_tcc -name "*counter_collection.csv" | head -1)
{ echo "LDS / L2 counters, bench.py $PMCARGS on one stream lane, lib_sha16 $(sha256sum multi-modal-loam_amd/libmmloam_hip.so | cut -c1-16); sums over all dispatches of a kernel in the run (map build and checks included)"; echo; python tools/pmc_lds_l2.py $L $T; } > $OUT/${TAG}_lds_l2_counters$SUF.md
for f in $OUT/pmc_lds.err $OUT/pmc_tcc.err; do tail -n 3 $f | cut -c1-200; done
rm -rf $OUT/pmc_lds $OUT/pmc_tcc
head -30 $OUT/${TAG}_lds_l2_counters$SUF.md | cut -c1-180
