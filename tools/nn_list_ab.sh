# iba_nn_list_kernel (the persistent walk over the anchored lists, default) vs iba_nn_kernel's one block per slice (IBA_NN_LIST=0), same box, at the bench shape
export IBA_DEBUG_ENV=1
cd $GRAFT_REPO_ROOT
run() {
  env "$@" python3 bench.py --steps 20 --warmup 3 --settle 300 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); k = d['roofline']['kernel_ms']
print('$*', 'evals/s %.0f' % d['value'], 'ms/step %.4f' % d['ms_per_step'], {a: round(b, 4) for a, b in k.items()})"
}
for R in ${@:-1 0 1 0}; do run IBA_NN_LIST=$R; done
