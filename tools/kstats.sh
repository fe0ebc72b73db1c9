# per-kernel average durations of the evaluation chain at the bench shape (rocprofv3 --kernel-trace --stats on tools/pmc_probe.py full)
# usage (GPU box): bash tools/kstats.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-kstats}
mkdir -p gpurun_out/$T
export PMC_PROBE_ITERS=${ITERS:-30}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/st -o st -- python3 ${PROBE:-tools/pmc_probe.py} ${PROBE_ARG:-full} > /dev/null 2> gpurun_out/$T/log.txt
python3 - <<'PY' $T
import csv, glob, sys
for fn in glob.glob("gpurun_out/%s/st/**/*kernel_stats.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(fn)):
        print("%-60s calls %4s avg %9.1f us  total %6.1f%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
