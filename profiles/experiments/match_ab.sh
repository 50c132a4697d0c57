# match_ab.sh a.so b.so ...: the 11v11 bench (8 192 matches x 64 cycles, 16 launches per region) on each build, interleaved, 3 rounds.
# A name of the form general:lib.so runs that build's run-time-parameter instantiation (S2D_MATCH_GENERAL_KERNEL=1).
run() {
  lib=${1#general:}; gen=0; [ "$lib" != "$1" ] && gen=1
  S2D_MATCH_GENERAL_KERNEL=$gen S2D_LIB=$lib python bench.py --task match --steps 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), round(d['roofline']['launch_us'],1), [round(x/1e9,2) for x in d['repeats']])"
}
for r in 1 2 3; do for lib in "$@"; do run $lib; done; done
