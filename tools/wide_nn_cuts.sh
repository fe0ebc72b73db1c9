# phase cuts of the search kernel (no anchored lists near: every lane searches the tree) on a box-wide batch (spread x 40)
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for k in ${CUTS:-1 2 4 5 3 0}; do
  IBA_NN_DBG=$k python tools/split_probe.py 200 10000 40 2>&1 | grep "B=64 cost" | sed "s/^/nn_dbg=$k /"
done
