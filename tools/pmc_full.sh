# one rocprofv3 --pmc pass of the full chain (tools/pmc_probe.py): usage: bash tools/pmc_full.sh "<counters>" <tag> [mode]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$2
rocprofv3 --pmc $1 --output-format csv -d gpurun_out/$2 -o pmc -- python3 tools/pmc_probe.py ${3:-cost} > /dev/null 2> gpurun_out/$2/log.txt
python3 - $2 <<'PY'
import csv, collections, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in glob.glob(f"gpurun_out/{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(k, {c: "%.4g" % (acc[k][c] / n[k][c]) for c in sorted(acc[k])})
PY
