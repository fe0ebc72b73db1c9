# counters of one kernel of the evaluation chain (tools/pmc_probe.py full). usage (GPU box): bash tools/pmc_kernel.sh <tag> <kernel name substring>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-pk}; K=${2:-iba_pairs}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); d=gpurun_out/$T/p$i; mkdir -p $d
  rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- python3 tools/pmc_probe.py full > /dev/null 2> $d/log.txt
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
