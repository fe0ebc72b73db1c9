# the derivative half of the candidates carried by the association kernel's spare blocks (default) vs a copy launch of its own (IBA_JETS_FOLD=0), same box
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for R in 1 0 1 0; do
  IBA_JETS_FOLD=$R python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('JETS_FOLD=$R', 'evals/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], k)"
done
