# match_pmc.sh <outdir>: the two SQ counter passes of the 11v11 rollout kernel (instruction mix, wait/busy cycles), then the summary
OUT=$1; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
MIX1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"
MIX2="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $MIX1 --output-format csv -d $OUT/pmc_match_mix1 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/pmc_match_mix1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc $MIX2 --output-format csv -d $OUT/pmc_match_mix2 -- python3 bench.py --task match --steps 8 --warmup 1 --no-cpu-baseline > $OUT/pmc_match_mix2.log 2>&1 &&
python3 profiles/summarise_pmc.py $OUT > $OUT/summary.txt 2>&1; tail -20 $OUT/summary.txt
