cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
SARA_HIP_STREAMS=1 python bench.py --cpu-frames 0 > $D/r01b_bench_single_stream.json 2>> $D/r01b_bench.err
SARA_HIP_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r01b_ss -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 > $D/r01b_ss.log 2>&1
python bench.py --cpu-frames 0 > $D/r01b_bench_multi_again.json
