#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag>
# Produces gpurun_out/<tag>/ : the default bench line, per-mode lines, rocprofv3 kernel stats of the SAME bench command as the
# default line, PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and instruction-mix passes.  Copy what is to be judged
# into profiles/<round>/ (gpurun_out/ is scratch).
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --fuse 64 --steps 64 --no-secondary --no-cpu-baseline > $OUT/bench_rollout_T64.json 2> /dev/null
python bench.py --noise --no-secondary --no-cpu-baseline > $OUT/bench_rollout_noise.json 2> /dev/null
python bench.py --mode step --no-cpu-baseline > $OUT/bench_step.json 2> /dev/null
python bench.py --envs 4096 --no-secondary --no-cpu-baseline > $OUT/bench_rollout_4096.json 2> /dev/null
python bench.py --envs 1048576 --fuse 64 --steps 16 --no-secondary --no-cpu-baseline > $OUT/bench_rollout_1M.json 2> /dev/null
python bench.py --task match --steps 16 > $OUT/bench_match_rollout.json 2> /dev/null
python bench.py --task match --mode step --steps 512 --no-cpu-baseline > $OUT/bench_match_step.json 2> /dev/null
# kernel stats of the same commands (no PMC in these runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rollout -- python3 bench.py --no-cpu-baseline --no-secondary > $OUT/prof_rollout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_step -- python3 bench.py --mode step --no-cpu-baseline > $OUT/prof_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_match -- python3 bench.py --task match --steps 16 --no-cpu-baseline > $OUT/prof_match.log 2>&1
for d in prof_rollout prof_step prof_match; do find $OUT/$d -name "*kernel_stats.csv" | while read f; do echo "== $d"; head -6 "$f"; done; done > $OUT/kernel_stats_summary.txt
# PMC passes: traffic (each counter its own pass), then instruction mix
pmc() { NAME=$1; CTR=$2; shift 2; rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/pmc_${NAME} -- python3 bench.py --no-cpu-baseline --no-secondary "$@" > $OUT/pmc_${NAME}.log 2>&1; }
pmc rollout256_fetch FETCH_SIZE --steps 8 --warmup 1
pmc rollout256_write WRITE_SIZE --steps 8 --warmup 1
pmc rollout64rot_fetch FETCH_SIZE --fuse 64 --steps 18 --warmup 3
pmc rollout64rot_write WRITE_SIZE --fuse 64 --steps 18 --warmup 3
pmc rollout64one_fetch FETCH_SIZE --fuse 64 --steps 16 --warmup 1 --rotate-buffers 1
pmc rollout64one_write WRITE_SIZE --fuse 64 --steps 16 --warmup 1 --rotate-buffers 1
pmc noise256_fetch FETCH_SIZE --noise --steps 8 --warmup 1
pmc noise256_write WRITE_SIZE --noise --steps 8 --warmup 1
pmc r4096_fetch FETCH_SIZE --envs 4096 --steps 16 --warmup 2
pmc r4096_write WRITE_SIZE --envs 4096 --steps 16 --warmup 2
pmc r1m_fetch FETCH_SIZE --envs 1048576 --fuse 64 --steps 4 --warmup 1
pmc r1m_write WRITE_SIZE --envs 1048576 --fuse 64 --steps 4 --warmup 1
pmc step_fetch FETCH_SIZE --mode step --steps 1024 --warmup 64
pmc step_write WRITE_SIZE --mode step --steps 1024 --warmup 64
pmc match_fetch FETCH_SIZE --task match --steps 8 --warmup 1
pmc match_write WRITE_SIZE --task match --steps 8 --warmup 1
MIX1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
MIX2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $MIX1 --output-format csv -d $OUT/pmc_rollout_mix1 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/pmc_rollout_mix1.log 2>&1
rocprofv3 --kernel-trace --pmc $MIX2 --output-format csv -d $OUT/pmc_rollout_mix2 -- python3 bench.py --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/pmc_rollout_mix2.log 2>&1
rocprofv3 --kernel-trace --pmc $MIX1 --output-format csv -d $OUT/pmc_noise_mix1 -- python3 bench.py --noise --no-cpu-baseline --no-secondary --steps 8 --warmup 1 > $OUT/pmc_noise_mix1.log 2>&1
rocprofv3 --kernel-trace --pmc $MIX1 --output-format csv -d $OUT/pmc_match_mix1 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/pmc_match_mix1.log 2>&1
rocprofv3 --kernel-trace --pmc $MIX2 --output-format csv -d $OUT/pmc_match_mix2 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/pmc_match_mix2.log 2>&1
python3 profiles/summarise_pmc.py $OUT
