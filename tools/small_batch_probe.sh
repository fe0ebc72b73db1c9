# one-candidate and small-batch calls (cost tuple; cost + normal equations; frozen factors), from C, with the shared pair search from 1 candidate up (default)
# and only from 5 / 15 candidates up (IBA_COMMON_MIN_BATCH: smaller batches take the one-launch per-candidate association)
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for M in 1 5 15; do echo "IBA_COMMON_MIN_BATCH=$M"; IBA_COMMON_MIN_BATCH=$M python3 tools/latency_probe.py 2>&1 | grep -v "^$"; done
