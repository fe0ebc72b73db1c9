# factor kernel with the one-line gather records (default) vs the separate arrays (IBA_FACTOR_REC=0), same box: factor + sums time at the bench shape
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for R in 1 0 1 0; do
  IBA_FACTOR_REC=$R python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('REC=$R', 'evals/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'factor+sums %.4f ms' % k['factor + sums'])"
done
