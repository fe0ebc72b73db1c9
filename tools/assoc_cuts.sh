cd $GRAFT_REPO_ROOT
for k in 1 2 3 4 5 6 7 0; do
  IBA_ASSOC_DBG=$k python tools/split_probe.py 2>&1 | grep "B=64 cost" | sed "s/^/dbg=$k /"
done
