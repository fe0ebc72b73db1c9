# cumulative time of iba_assoc2_kernel's phases at the bench shape (IBA_ASSOC_DBG=k ends the kernel after phase k: 1 tables, 4 pair pass,
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
# 5 ties, 6 list, 7 entries, 0 everything) and of iba_nn_kernel's (IBA_NN_DBG: 1 staging, 2 -, 4 direct pass without tree searches, 5 searches without finish, 3 before the sums)
cd $GRAFT_REPO_ROOT
for k in 1 4 5 6 7 0; do IBA_ASSOC_DBG=$k python tools/split_probe64.py 2>&1 | grep " full" | sed "s/^/assoc_dbg=$k /"; done
for k in 1 2 4 5 3 0; do IBA_NN_DBG=$k python tools/split_probe64.py 2>&1 | grep " full" | sed "s/^/nn_dbg=$k /"; done
