for r in 1 2; do for L in "$@"; do
  S2D_LIB=$L S2D_ROLLOUT_WS=0 python bench.py --steps 16 --warmup 2 --envs 1048576 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], round(d['value']/1e9,2), 'G steps/s', round(d['roofline']['launch_us'],1), 'us', d['roofline']['kernel'])"
done; done
