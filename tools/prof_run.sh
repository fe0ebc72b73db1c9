# the rocprofv3 passes behind profiles/<tag>_rocprof_summary.md and profiles/pmc_latest.json, all on bench.py itself:
#   kernel trace + stats, then one --pmc pass per counter list (no tracing together with counters).
# usage (GPU box): bash tools/prof_run.sh <tag>       then, back in the build container: python tools/summarize_prof.py <tag>
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-r03}
mkdir -p gpurun_out/$T
BENCH="python3 bench.py --steps 15 --warmup 2 --settle 100 --no-cpu-baseline --no-extras"   # (--settle 100: the profiler does not need the clocks at their sustained state, and 2500 untimed steps x 6 passes are 100 MB of traces)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/stats -o stats -- $BENCH > gpurun_out/$T/bench_under_rocprof.json 2> gpurun_out/$T/rocprof_stats.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$T/fetch -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$T/write -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_write.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES --output-format csv -d gpurun_out/$T/insts -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_insts.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$T/cycles -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_cycles.log
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$T/l2 -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_l2.log   # (r06: the L2 request rate — what bounds the association and factor kernels)
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/$T/mfma -o pmc -- $BENCH > /dev/null 2> gpurun_out/$T/pmc_mfma.log
python3 bench.py --steps 20 --warmup 3 > gpurun_out/$T/bench_full.json 2> gpurun_out/$T/bench_full.err
ls gpurun_out/$T
