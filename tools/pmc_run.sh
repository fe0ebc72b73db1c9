# usage: bash tools/pmc_run.sh <tag> "<counter list>"   -> gpurun_out/<tag>/*.csv (one pass per invocation)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$1
rocprofv3 --pmc $2 --output-format csv -d gpurun_out/$1 -o pmc -- python3 tools/pmc_probe.py full > /dev/null 2> gpurun_out/$1/log.txt
ls gpurun_out/$1
