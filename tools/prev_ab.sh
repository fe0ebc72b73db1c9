# the library in the tree against a build of an earlier tree (libiba_prev.so: `git stash; make -C csrc OUT=../libiba_prev.so; git stash pop; make -C csrc`), same box, bench shape
export IBA_DEBUG_ENV=1
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/spatial-temporal-lidar-camera-calibration_amd
run() {
  env "$@" python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('$*', 'evals/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], {a: round(b, 4) for a, b in k.items()})"
}
for R in 1 2 3; do run A=new; run IBA_LIB=$P/libiba_prev.so; done
