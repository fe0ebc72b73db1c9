# instruction-cache counters of one kernel at a given scene shape. usage (GPU box): bash tools/pmc_icache.sh <tag> <kernel substring> <frames> <points per frame>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-ic}; K=${2:-iba_nn}; export PMC_FRAMES=${3:-200}; export PMC_PTS=${4:-10000}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); d=gpurun_out/$T/p$i; mkdir -p $d
  rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python3 tools/pmc_probe.py cost > /dev/null 2> $d/log.txt
done
python3 - <<'PY' $T $K
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for fn in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(fn)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()): print("   %-28s %.4g" % (c, sum(v) / len(v)))
PY
rm -rf gpurun_out/$T
