# per-phase instruction counters of the two-kernel path: one rocprofv3 --pmc pass per cut point (IBA_ASSOC_DBG / IBA_NN_DBG cut the
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
# kernels short after a phase; results are garbage, counters are not). usage: bash tools/pmc_split.sh "<counters>" <tag> [mode]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mode=${3:-cost}
for k in a1 a2 a3 a4 a5 a6 a7 n1 n2 n4 n5 full; do
  mkdir -p gpurun_out/$2/$k
  unset IBA_ASSOC_DBG IBA_NN_DBG
  case $k in a*) export IBA_ASSOC_DBG=${k#a};; n*) export IBA_NN_DBG=${k#n};; esac
  rocprofv3 --pmc $1 --output-format csv -d gpurun_out/$2/$k -o pmc -- python3 tools/pmc_probe.py $mode > /dev/null 2> gpurun_out/$2/$k/log.txt
done
unset IBA_ASSOC_DBG IBA_NN_DBG
python3 tools/pmc_split_summary.py $2 > gpurun_out/$2/summary.txt 2>&1
cat gpurun_out/$2/summary.txt
