# phase cuts of the per-candidate association kernel (iba_assoc_kernel) on a batch as wide as the search box (spread x 200)
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for k in ${CUTS:-1 2 3 4 5 6 7 0}; do
  IBA_ASSOC_DBG=$k python tools/split_probe.py 200 10000 200 2>&1 | grep "B=64 cost" | sed "s/^/dbg=$k /"
done
