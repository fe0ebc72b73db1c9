# per-kernel times at the reference's scan size on the full 200 keyframes (24 M points): rocprofv3 --kernel-trace --stats on tools/kitti_probe.py
# usage (GPU box): bash tools/kitti_prof.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-kitti}
mkdir -p gpurun_out/$T
KITTI_TILE=5 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/stats -o stats -- python3 tools/kitti_probe.py 20 > gpurun_out/$T/probe.txt 2> gpurun_out/$T/rocprof.log
cat gpurun_out/$T/probe.txt
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/$T/stats/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows[:12]:
    print("%-60s calls %5s avg %9.1f us  %s %%" % (r["Name"].replace("void ", "").replace("iba::", "").split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
