# Copies the judged summaries of a collect_profiles.sh run from gpurun_out/prof into profiles/.
#   bash tools/copy_profiles.sh <tag>
T=$1; S=gpurun_out/prof; D=profiles
cp $S/${T}_bench.json $D/${T}_bench.json
cp $S/${T}_bench_single_stream.json $D/${T}_bench_single_stream.json
cp $S/${T}_ms_kernel_stats.csv $D/${T}_kernel_stats.csv
cp $S/${T}_ss_kernel_stats.csv $D/${T}_kernel_stats_single_stream.csv
cp $S/${T}_4k_kernel_stats.csv $D/${T}_4k_kernel_stats_single_stream.csv
cp $S/${T}_match_mfma_kernel_stats.csv $S/${T}_match_exh_kernel_stats.csv $D/ 2>/dev/null
mv $D/${T}_match_exh_kernel_stats.csv $D/${T}_match_exhaustive_kernel_stats.csv 2>/dev/null
for f in match_probe.txt pmc_fetch_size.csv pmc_write_size.csv pmc_traffic.txt spans.txt sq_counters.txt \
         single_frame_timeline.txt single_frame_unprofiled.txt thread_calls.txt blur_fma_probe.txt lds_counters.txt; do
  [ -f $S/${T}_$f ] && cp $S/${T}_$f $D/${T}_$f
done
[ -f $S/pyramid_traffic.json ] && cp $S/pyramid_traffic.json $D/pyramid_traffic.json
ls $D | grep "^${T}_" | wc -l
