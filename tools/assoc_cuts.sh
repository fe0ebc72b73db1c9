# cumulative time of the association kernel's phases (IBA_ASSOC_DBG cuts the kernel short; results are garbage, times are not)
cd $GRAFT_REPO_ROOT
for k in ${CUTS:-1 2 3 4 5 6 7 0}; do
  IBA_ASSOC_DBG=$k python tools/split_probe.py 2>&1 | grep "B=64 cost" | sed "s/^/dbg=$k /"
done
