# factor kernel time at the bench shape by ranges per candidate (IBA_FACTOR_WAVES_PER_CAND) (IBA_FACTOR_V2=1) and for the default one-wave-per-keyframe kernel
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
# usage (GPU box): bash tools/factor_sweep.sh "<list of W>"
cd $GRAFT_REPO_ROOT
for W in ${1:-0 16 32 48 64 100 200}; do
  IBA_FACTOR_V2=1 IBA_FACTOR_WAVES_PER_CAND=$W python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('W=$W', 'evals/s %.0f' % d['value'], 'factor+sums %.4f ms' % k['factor + sums'])"
done
IBA_FACTOR_V2=0 python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('V1', 'evals/s %.0f' % d['value'], 'factor+sums %.4f ms' % k['factor + sums'])"
