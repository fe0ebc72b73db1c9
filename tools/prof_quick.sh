# a short form of tools/prof_run.sh: kernel trace + the instruction and cycle counters only (three passes), into gpurun_out/<tag>/; then python tools/summarize_prof.py <tag>
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-quick}
mkdir -p gpurun_out/$T
BENCH="python3 bench.py --steps 15 --warmup 2 --settle 100 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/stats -o stats -- $BENCH > gpurun_out/$T/bench_under_rocprof.json 2> gpurun_out/$T/rocprof_stats.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES --output-format csv -d gpurun_out/$T/insts -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_insts.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/cycles -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_cycles.log
ls gpurun_out/$T
