#!/bin/bash
# A/B of the workgroup size (timing-only builds libs2d_hip_b{64,128,512}.so next to the default 256)
for B in 64 128 256 512; do
  L=gym-soccer-2d-env_amd/lib/libs2d_hip_b$B.so; [ $B = 256 ] && L=gym-soccer-2d-env_amd/lib/libs2d_hip.so
  echo "== block $B"; S2D_LIB=$L python bench.py --steps 32 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value']/1e9, 'G steps/s', d['roofline']['launch_us'], 'us/launch')"
  S2D_LIB=$L python bench.py --steps 32 --warmup 2 --no-cpu-baseline --envs 1048576 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('1M:', d['value']/1e9, 'G steps/s')"
done
