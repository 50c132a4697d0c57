run() { python bench.py --task match --steps 16 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e9,3), round(d['roofline']['launch_us'],1), [round(x/1e9,2) for x in d['repeats']], d['events'])"; }
run default
S2D_MATCH_KW='{"stopped_clock":0}' run clock_runs
S2D_MATCH_KW='{"stopped_clock":0,"announce_wait":0}' run clock_runs_wait0
S2D_MATCH_KW='{"after_goal_wait":0,"announce_wait":0}' run nowaits
