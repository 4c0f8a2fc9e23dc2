#!/bin/bash
# A/B matrix of mml_step's schedules on the configs[1] bench (short runs): gpurun_out/<tag>/matrix.txt
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {  # label, env...
  L=$1; shift
  env "$@" python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --skip-upload --kernel-steps 1 $EXTRA > $OUT/m_$L.json 2> $OUT/m_$L.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/m_$L.json").read().strip().splitlines()[-1])
    print("%-28s %9.0f scans/s  %7.2f ms/step  replica %s" % ("$L", r["value"], r["ms_per_step"], r.get("replica_check", {}).get("mismatches")))
except Exception as e:
    print("%-28s failed: %s" % ("$L", e))
PY
}
for spec in "$@"; do
  L=$(echo "$spec" | tr ' =/' '___')
  run "$L" $spec
done | tee $OUT/matrix.txt
