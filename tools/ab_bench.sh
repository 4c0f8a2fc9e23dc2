#!/bin/bash
# Same-box A/B of two builds on bench.py's own workload (64 distinct scans on the tiled map: the association stages see far
# queries the one-tile map of tools/stage_probe.py does not produce):  bash tools/ab_bench.sh <prev .so> [new .so] [bench args]
#   make -C multi-modal-loam_amd/csrc BUILD=build_prev OUT=../libmmloam_hip_prev.so   (at the commit to compare against)
PREV=${1:-multi-modal-loam_amd/libmmloam_hip_prev.so}; NEW=${2:-multi-modal-loam_amd/libmmloam_hip.so}; shift 2 2>/dev/null
for i in 1 2; do
for L in $PREV $NEW; do
MML_LIB_PATH=$L python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-upload "$@" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=r['roofline']['stage_ms_per_launch']
print('$L'.split('/')[-1], round(r['value']), 'sum %.3f'%sum(st.values()), ' '.join('%s %.3f'%(k,v) for k,v in st.items()), 'replica mismatches', r['replica_check']['mismatches'], r['errors'])"
done; done
