# timing cuts of iba_factor2_kernel at the bench shape (IBA_FACTOR_DBG: bit 0 skips the plane-factor bodies, bit 1 the 3d-3d bodies; results invalid)
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for D in ${1:-0 1 2 3}; do
  IBA_FACTOR_V2=1 IBA_FACTOR_DBG=$D python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('DBG=$D', 'factor+sums %.4f ms' % k['factor + sums'])"
done
