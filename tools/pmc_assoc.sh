# what the association kernel waits for: LDS / instruction-fetch / VMEM counters in separate --pmc passes on bench.py. usage: bash tools/pmc_assoc.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras"
i=0
for set in "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ATOMIC_RETURN SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"; do
  i=$((i+1)); d=gpurun_out/$1/p$i; mkdir -p $d
  rocprofv3 --pmc $set --output-format csv -d $d -o pmc -- $BENCH > /dev/null 2> $d/log.txt
done
python3 - <<'PY' $1
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iba::", "")
        if k.startswith(("iba_assoc", "iba_nn_kernel", "iba_factor_kernel")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()): print("   %-28s %.4g" % (c, sum(v) / len(v)))
PY
