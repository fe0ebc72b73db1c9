# per-phase instruction counters of the fused frame kernel: one rocprofv3 --pmc pass per build cut short after a phase
# (libiba_exp_stop<k>.so built with -DIBA_STOP_AFTER=k) plus the full kernel. usage: bash tools/pmc_phases.sh "<counters>" <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 0 1 7 2 3 4 5 6 full; do
  mkdir -p gpurun_out/$2/$k
  if [ $k = full ]; then unset IBA_LIB; else export IBA_LIB=$PWD/spatial-temporal-lidar-camera-calibration_amd/libiba_exp_stop$k.so; fi
  rocprofv3 --pmc $1 --output-format csv -d gpurun_out/$2/$k -o pmc -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/$2/$k/log.txt
done
ls gpurun_out/$2/*
