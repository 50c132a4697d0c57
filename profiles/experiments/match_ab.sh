# match_ab.sh a.so b.so ...: the 11v11 bench (8 192 matches x 64 cycles, 16 launches per region) on each build, interleaved, 3 rounds
run() { S2D_LIB=$1 python bench.py --task match --steps 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), round(d['roofline']['launch_us'],1), [round(x/1e9,2) for x in d['repeats']], d['events'])"; }
for r in 1 2 3; do for lib in "$@"; do run $lib; done; done
