OUT=gpurun_out/r04final
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rollout -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/prof_rollout.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -- python3 bench.py --mode step --no-cpu-baseline > $OUT/prof_step.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -- python3 bench.py --task match --steps 16 --no-cpu-baseline > $OUT/prof_match.log 2>&1 &&
for d in prof_rollout prof_step prof_match; do find $OUT/$d -name "*kernel_stats.csv" | while read f; do echo "== $d"; head -6 "$f"; done; done > $OUT/kernel_stats_summary.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
cat $OUT/kernel_stats_summary.txt
