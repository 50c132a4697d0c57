#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag>
# Produces gpurun_out/<tag>/ : bench lines for the three modes and rocprofv3 kernel stats
# of the SAME bench command as the default (rollout) line.
set -e
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 64 --warmup 4 > $OUT/bench_rollout.json 2> $OUT/bench_rollout.err
python bench.py --steps 2048 --warmup 128 --mode step --no-cpu-baseline > $OUT/bench_step.json 2> $OUT/bench_step.err
python bench.py --steps 32 --warmup 2 --mode graph --no-cpu-baseline > $OUT/bench_graph.json 2> $OUT/bench_graph.err
python bench.py --steps 64 --warmup 4 --envs 1048576 --no-cpu-baseline > $OUT/bench_rollout_1M.json 2> $OUT/bench_rollout_1M.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rollout -- python3 bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-secondary > $OUT/prof_rollout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -- python3 bench.py --steps 2048 --warmup 128 --mode step --no-cpu-baseline > $OUT/prof_step.log 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 "$f"; done > $OUT/kernel_stats_summary.txt
cat $OUT/bench_*.json
cat $OUT/kernel_stats_summary.txt
# 11v11 match engine (BASELINE.json configs[3])
python bench.py --task match --steps 16 --warmup 1 > $OUT/bench_match_rollout.json 2> $OUT/bench_match_rollout.err
python bench.py --task match --steps 512 --warmup 32 --mode step --no-cpu-baseline > $OUT/bench_match_step.json 2> $OUT/bench_match_step.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -- python3 bench.py --task match --steps 16 --warmup 1 --no-cpu-baseline > $OUT/prof_match.log 2>&1
find $OUT/prof_match -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -5 "$f"; done >> $OUT/kernel_stats_summary.txt
