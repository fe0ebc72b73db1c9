# counters of iba_plane_kernel at the bench shape: separate --pmc passes on tools/plane_probe.py. usage: bash tools/pmc_plane.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  d=gpurun_out/$1/$(echo $set | cut -d' ' -f1)
  mkdir -p $d
  rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python3 tools/plane_probe.py 200 10000 memo > /dev/null 2> $d/log.txt
done
python3 - <<'PY' $1
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iba::", "")
        if "plane" in k or "fit" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
