# same-box A/B of the undistortion fork and the lane count on the bench workload:  bash tools/ab_env_fork.sh
for i in 1 2; do for L in 1 2 3; do for F in 0 1; do
MML_LANES=$L MML_UND_FORK=$F python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --skip-upload 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lanes=$L fork=$F', round(r['value']), 'ms/step %.2f'%r['ms_per_step'], 'replica mismatches', r['replica_check']['mismatches'], r['errors'])"
done; done; done
