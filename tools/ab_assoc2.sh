cd $GRAFT_REPO_ROOT
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
for cv in 3 12; do
for lib in libiba_base.so libiba_mi355x.so libiba_w8.so; do
  PROBE_COVIS=$cv IBA_LIB=$PWD/spatial-temporal-lidar-camera-calibration_amd/$lib python tools/split_probe.py 2>&1 | grep "B=64" | sed "s/^/covis=$cv $lib /"
done; done
