# LDS-pipe counters of the per-keypoint kernels of one bench step (single stream).
#   bash tools/lds_counters.sh <tag> [lib.so]   -> gpurun_out/prof/<tag>_lds_counters.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-lds}
[ -n "$2" ] && export SARA_HIP_SIFT_LIB=$2
mkdir -p $R/gpurun_out/prof
cd $R
bash tools/pmc_pipeline.sh ${TAG}_lds "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU" > /dev/null
( echo "# bench.py step, single stream, --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU; per-launch means, M = 1e6; lib = ${SARA_HIP_SIFT_LIB:-default}"
  python tools/pmc_raw.py $R/gpurun_out/pmc_pipe/${TAG}_lds_counter_collection.csv | grep "descriptor_kernel\|orientation_kernel" ) > $R/gpurun_out/prof/${TAG}_lds_counters.txt
cat $R/gpurun_out/prof/${TAG}_lds_counters.txt
