# same-box A/B of library builds at the reference's scan size (40 KF x 120 k points): tools/ab_kitti.sh libA.so libB.so
export IBA_DEBUG_ENV=1   # the library reads its environment overrides only with this set (round 6)
cd $GRAFT_REPO_ROOT
for lib in "$@"; do IBA_LIB=$PWD/spatial-temporal-lidar-camera-calibration_amd/$lib python tools/split_probe.py 40 120000 2>&1 | grep "B=64" | sed "s/^/$lib /"; done
