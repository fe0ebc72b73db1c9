# instruction counts of the association kernel phase by phase: the kernel cut short after each phase (IBA_ASSOC_DBG), one
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
# --pmc pass per cut. usage (GPU box): bash tools/pmc_cuts.sh <tag>   -> gpurun_out/<tag>/cuts.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-cuts}
mkdir -p gpurun_out/$T
for k in ${CUTS:-1 2 3 4 5 6 7 0}; do
  export IBA_ASSOC_DBG=$k
  d=gpurun_out/$T/c$k; mkdir -p $d
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d -o pmc -- python3 ${PROBE:-tools/pmc_probe.py} ${PROBE_ARG:-full} > /dev/null 2> $d/log.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/st -o st -- python3 ${PROBE:-tools/pmc_probe.py} ${PROBE_ARG:-full} > /dev/null 2>> $d/log.txt
done
unset IBA_ASSOC_DBG
python3 - <<'PY' $T > gpurun_out/$T/cuts.txt
import csv, glob, sys
from collections import defaultdict
T = sys.argv[1]
print("cut  (cumulative, per launch of 64 candidates x 200 frames)")
for k in "1 2 3 4 5 6 7 0".split():
    acc = defaultdict(list)
    for fn in glob.glob("gpurun_out/%s/c%s/**/*counter_collection.csv" % (T, k), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "iba_assoc" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = None
    for fn in glob.glob("gpurun_out/%s/c%s/st/**/*kernel_stats.csv" % (T, k), recursive=True):
        for r in csv.DictReader(open(fn)):
            if "iba_assoc" in r["Name"]:
                t = float(r["AverageNs"]) / 1e3
    print("dbg=%s  time_us=%s  " % (k, "%.1f" % t if t else "?") + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc.items())))
PY
cat gpurun_out/$T/cuts.txt
