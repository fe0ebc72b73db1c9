# factor kernel time at the bench shape by ranges per candidate (IBA_FACTOR_WAVES_PER_CAND) and for the one-wave-per-keyframe kernel (IBA_FACTOR_V1=1)
# usage (GPU box): bash tools/factor_sweep.sh "<list of W>"
cd $GRAFT_REPO_ROOT
for W in ${1:-0 16 32 48 64 100 200}; do
  IBA_FACTOR_WAVES_PER_CAND=$W python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('W=$W', 'evals/s %.0f' % d['value'], 'factor+sums %.4f ms' % k['factor + sums'])"
done
IBA_FACTOR_V1=1 python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('V1', 'evals/s %.0f' % d['value'], 'factor+sums %.4f ms' % k['factor + sums'])"
