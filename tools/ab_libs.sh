# same-box A/B of library builds: tools/ab_libs.sh libA.so libB.so ... (names under the package directory), two rounds each
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for lib in "$@"; do
  IBA_LIB=$PWD/spatial-temporal-lidar-camera-calibration_amd/$lib python tools/split_probe64.py 2>&1 | tail -2 | sed "s/^/$lib /"
done; done
