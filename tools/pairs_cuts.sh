# where iba_pairs_kernel's time goes at the bench shape: builds with -DIBA_PAIRS_CUT=k (libiba_cut<k>.so) end the kernel behind its k-th phase
# (1 chunk tests, 2 batch bound, 3 per-point window + "does any point walk", 4 keypoint grid staged in LDS, 5 walk; the regular library: + reservation and write-out);
# rocprofv3 kernel trace of a short bench run per build. Results of the cut builds are invalid; times only.
# build: make -C spatial-temporal-lidar-camera-calibration_amd/csrc OUT=../libiba_cut1.so CXXFLAGS="... -DIBA_PAIRS_CUT=1"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp IBA_DEBUG_ENV=1
P=$GRAFT_REPO_ROOT/spatial-temporal-lidar-camera-calibration_amd
mkdir -p gpurun_out/pairs_cuts
for L in libiba_cut1.so libiba_cut2.so libiba_cut3.so libiba_cut4.so libiba_cut5.so libiba_mi355x.so; do
  [ -f $P/$L ] || continue
  export IBA_LIB=$P/$L
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pairs_cuts/$L -o stats -- python3 bench.py --steps 10 --warmup 2 --settle 50 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/pairs_cuts/$L.log
  python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pairs_cuts/$L/**/*kernel_stats.csv", recursive=True)[0])))
for r in rows:
    if "iba_pairs_kernel" in r["Name"]: print("$L", "iba_pairs_kernel avg %.1f us over %s calls" % (float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
done
