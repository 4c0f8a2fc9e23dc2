#!/bin/bash
# Randomised parity campaign (tests/gpu_fuzz.py) over fresh seeds on the library as built: gpurun_out/<tag>/fuzz_campaign.txt
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
SHA=$(sha256sum multi-modal-loam_amd/libmmloam_hip.so | cut -c1-16)
ARGS="--lines 2500 --scans 80 --poses 40 --windows 90 --solves 30 --dense 4 --cubes 2 --maps 2 --batch 42 --dense-batch 60"
{
echo "Randomised parity campaign (tests/gpu_fuzz.py; device through the C-ABI against the CPU oracle) on lib_sha16 $SHA:"
echo "python tests/gpu_fuzz.py --seed S $ARGS"
for S in "$@"; do
  echo "== seed $S"
  python tests/gpu_fuzz.py --seed $S $ARGS 2>&1 | grep -v amdgpu.ids
  echo "rc ${PIPESTATUS[0]}"
done
} > $OUT/fuzz_campaign.txt
tail -15 $OUT/fuzz_campaign.txt
