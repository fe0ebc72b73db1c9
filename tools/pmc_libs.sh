# one rocprofv3 --pmc pass per experimental library. usage: bash tools/pmc_libs.sh "<counters>" <tag> <lib suffix> ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C="$1"; T="$2"; shift 2
for k in "$@"; do
  mkdir -p gpurun_out/$T/$k
  export IBA_LIB=$PWD/spatial-temporal-lidar-camera-calibration_amd/libiba_exp_$k.so
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/$T/$k -o pmc -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/$T/$k/log.txt
  python3 - <<PY
import csv,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open("gpurun_out/$T/$k/pmc_counter_collection.csv")):
    if "frame_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
print("$k", {c: "%.4g" % (acc[c]/n[c]) for c in sorted(acc)})
PY
done
