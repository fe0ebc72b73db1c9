# kernel timeline of one small-batch evaluation chain: durations and the gaps between consecutive kernels
# usage (GPU box): bash tools/chain_gaps.sh <B> <cost|full|factors> [frames of the shard, default 200]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B=${1:-1}; M=${2:-cost}; FR=${3:-200}
mkdir -p gpurun_out/gaps
cat > /tmp/gaps_run.py <<PY
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
PKG = "spatial-temporal-lidar-camera-calibration_amd"
pkg = importlib.import_module(PKG); synth = importlib.import_module(PKG + ".synth"); abi = importlib.import_module(PKG + ".abi")
prob, meta = synth.make_scene(n_frames=200, pts_per_frame=10000, seed=0)
h = pkg.IbaHandle(prob, abi.reference_yaml_params(), frame_begin=0, frame_end=$FR)
h.build_problem(meta["x_gt"])
xs = synth.perturb(meta["x_gt"], np.random.default_rng(0), n=$B)
fn = {"cost": h.eval_cost, "full": h.eval_full, "factors": h.eval_factors}["$M"]
for _ in range(60): fn(xs)
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps/t -o t -- python3 /tmp/gaps_run.py > /dev/null 2> gpurun_out/gaps/log.txt
python3 - <<'PY'
import csv, glob
rows = []
for fn in glob.glob("gpurun_out/gaps/t/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-120:]
prev = None
for r in rows[-24:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s dur %7.1f us   gap before %7.1f us" % (r["Kernel_Name"].split("(")[0][-40:], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
PY
