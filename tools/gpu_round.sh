#!/bin/bash
# One GPU call of a build round: a subset of the GPU suite, then the bench lines, all logs under gpurun_out/<tag>/.
#   bash tools/gpu_round.sh <tag> "<pytest args>" [bench configs, e.g. "1 2"]
TAG=$1; TESTS=$2; CONFIGS=${3:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$TESTS" ]; then
  python -m pytest $TESTS -m gpu -x -q -s > $OUT/tests.log 2>&1
  echo "pytest rc=$?" >> $OUT/tests.log
  tail -8 $OUT/tests.log
fi
for c in $CONFIGS; do
  SUF=""; [ "$c" != "1" ] && SUF=_config$c
  python bench.py --config $c --strict > $OUT/bench$SUF.json 2> $OUT/bench$SUF.err
  echo "bench --config $c rc=$?" | tee -a $OUT/bench$SUF.err
  tail -3 $OUT/bench$SUF.err
  python - <<PY
import json
try:
    r = json.loads(open("$OUT/bench$SUF.json").read().strip().splitlines()[-1])
    print("value", r["value"], r["unit"], "ms_per_step", r["ms_per_step"])
    rf = r.get("roofline", {})
    print("roofline", rf.get("kernel"), rf.get("frac"), "stage_ms", {k: round(v, 4) for k, v in (rf.get("stage_ms_per_launch") or {}).items()})
    print("replica_check", r.get("replica_check"), "errors", r.get("errors"))
    if "full_window_imu" in r: print("full_window_imu", json.dumps(r["full_window_imu"])[:900])
except Exception as e:
    print("no bench line:", e)
PY
done
