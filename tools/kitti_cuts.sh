# phase cuts of the association and search kernels at the reference's scan size (40 keyframes x 120 k points, 64 candidates)
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for k in 0 1 4 5 6 7; do IBA_ASSOC_DBG=$k python tools/split_probe.py 40 120000 2>&1 | grep "B=64 cost" | sed "s/^/assoc_dbg=$k /"; done
for k in 1 2 4 5 3; do IBA_NN_DBG=$k python tools/split_probe.py 40 120000 2>&1 | grep "B=64 cost" | sed "s/^/nn_dbg=$k /"; done
python tools/kitti_probe.py 2>&1 | tail -3
