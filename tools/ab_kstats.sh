# A/B of two builds of the library on the same box: per-kernel averages (30 evaluations each), then the bench line of each
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
# usage (GPU box): bash tools/ab_kstats.sh build/lib_base.so
cd $GRAFT_REPO_ROOT
for L in "$1" ""; do
  if [ -n "$L" ]; then export IBA_LIB=$GRAFT_REPO_ROOT/$L; echo "== $L"; else unset IBA_LIB; echo "== in-tree"; fi
  bash tools/kstats.sh ab 2>&1 | grep -E "assoc|pairs|nn_kernel|factor|reduce2|fetch"
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: %.0f evals/s  %.4f ms/step' % (r['value'], r['ms_per_step']))"
done
