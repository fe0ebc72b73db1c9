set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01j
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01j/stats -o stats -- python3 bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r01j/bench_under_rocprof.json 2> gpurun_out/r01j/rocprof_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01j/fetch -o fetch -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/r01j/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01j/write -o write -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/r01j/pmc_write.log
ls gpurun_out/r01j/*
