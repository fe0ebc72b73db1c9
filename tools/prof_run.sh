set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01k
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01k/stats -o stats -- python3 bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r01k/bench_under_rocprof.json 2> gpurun_out/r01k/rocprof_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01k/fetch -o fetch -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/r01k/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01k/write -o write -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/r01k/pmc_write.log
ls gpurun_out/r01k/*
