# Collects the round's profile artefacts on the GPU box (outputs under gpurun_out/prof).
#   bash tools/collect_profiles.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
D=$R/gpurun_out/prof
mkdir -p $D
cd $R
python bench.py > $D/${TAG}_bench.json 2> $D/${TAG}_bench.err
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 python bench.py --cpu-frames 0 --no-extras > $D/${TAG}_bench_single_stream.json 2>> $D/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_ms -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/${TAG}_ms.log 2>&1
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_ss -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/${TAG}_ss.log 2>&1
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D -o f -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/f.log 2>&1
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D -o w -- python bench.py --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/w.log 2>&1
python tools/pmc_traffic.py $D 7 $D/pyramid_traffic.json > $D/${TAG}_pmc_traffic.txt
cp $D/f_counter_collection.csv $D/${TAG}_pmc_fetch_size.csv
cp $D/w_counter_collection.csv $D/${TAG}_pmc_write_size.csv
python tools/span_summary.py $D/${TAG}_ms_kernel_trace.csv > $D/${TAG}_spans.txt
bash tools/pmc_pipeline.sh ${TAG}_sq > /dev/null
python tools/pmc_pipeline_summary.py $R/gpurun_out/pmc_pipe/${TAG}_sq_counter_collection.csv > $D/${TAG}_sq_counters.txt
# 4K (config 5): kernel statistics of the same pipeline
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_4k -- python bench.py --width 3840 --height 2160 --octaves 5 --frames-per-gpu 16 --unique-frames 4 --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/${TAG}_4k.log 2>&1
# one 1080p frame per call: the launch timeline of a steady-state call
rocprofv3 --kernel-trace --output-format csv -d $D -o ${TAG}_b1 -- python tools/b1_probe.py > $D/${TAG}_b1.log 2>&1
( grep -v '^[WE]20' $D/${TAG}_b1.log; python tools/b1_timeline.py $D/${TAG}_b1_kernel_trace.csv ) > $D/${TAG}_single_frame_timeline.txt
# matcher kernels, the FMA-vs-exact blur probe, the thread probe
bash tools/match_profile.sh ${TAG} > $D/${TAG}_match.txt 2>&1
python tools/blur_fma_probe.py 2>/dev/null | grep -v "^ROCm\|^HIP \|^RCCL\|^Hostname\|^Librccl" > $D/${TAG}_blur_fma_probe.txt
for t in 1 2 4; do python tools/thread_calls.py $t 2>/dev/null | tail -1; done > $D/${TAG}_thread_calls.txt
python tools/b1_bench.py 2 2>/dev/null | grep "config2\|host time" > $D/${TAG}_single_frame_unprofiled.txt
ls $D | head -50
cat $D/${TAG}_spans.txt
