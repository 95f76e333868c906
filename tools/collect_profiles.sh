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
python tools/pmc_traffic.py $D 7 $D/pyramid_traffic.json
python tools/span_summary.py $D/${TAG}_ms_kernel_trace.csv > $D/${TAG}_spans.txt
bash tools/pmc_pipeline.sh ${TAG}_sq > /dev/null
python tools/pmc_pipeline_summary.py $R/gpurun_out/pmc_pipe/${TAG}_sq_counter_collection.csv > $D/${TAG}_sq_counters.txt
# 4K (config 5): kernel statistics of the same pipeline
SARA_HIP_STREAMS=1 SARA_HIP_SIDE_GRADIENT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ${TAG}_4k -- python bench.py --width 3840 --height 2160 --octaves 5 --frames-per-gpu 16 --unique-frames 4 --steps 5 --warmup 2 --cpu-frames 0 --no-extras > $D/${TAG}_4k.log 2>&1
# one 1080p frame per call: the launch timeline of a steady-state call
rocprofv3 --kernel-trace --output-format csv -d $D -o ${TAG}_b1 -- python tools/b1_probe.py > $D/${TAG}_b1.log 2>&1
( grep -v '^[WE]20' $D/${TAG}_b1.log; python tools/b1_timeline.py $D/${TAG}_b1_kernel_trace.csv ) > $D/${TAG}_single_frame_timeline.txt
ls $D | head -50
cat $D/${TAG}_spans.txt
