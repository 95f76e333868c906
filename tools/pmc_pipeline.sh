# Per-kernel SQ counters of one bench step (single stream): instruction mix and
# stall buckets.  Output: gpurun_out/pmc_pipe/*_counter_collection.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_pipe
cd $R
SARA_HIP_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_pipe -o sq -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $R/gpurun_out/pmc_pipe/sq.log 2>&1
tail -2 $R/gpurun_out/pmc_pipe/sq.log
