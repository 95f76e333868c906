# Per-kernel SQ counters of one bench step (single stream).
# usage: pmc_pipeline.sh <tag> "<counters>"   -> gpurun_out/pmc_pipe/<tag>_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-sq}
CTRS=${2:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY}
mkdir -p $R/gpurun_out/pmc_pipe
cd $R
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/pmc_pipe -o $TAG -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-extras > $R/gpurun_out/pmc_pipe/$TAG.log 2>&1
tail -1 $R/gpurun_out/pmc_pipe/$TAG.log
