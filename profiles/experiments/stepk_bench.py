"""s2d_step_k alone: us per launch / per cycle for k = 1, 2, 4, 8 (bench.py's measure_step_k).  S2D_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
dev = torch.device('cuda:0')
stream = torch.cuda.current_stream(dev)
eng = bench.reach_engine(65536, dev, 0, False)
eng.reset()
for k in (1, 2, 4, 8):
    m = bench.measure_step_k(eng, k, 2048 // k, 5, stream, 200.0)
    print(f"k={k}: {m['us_per_launch']:.2f} us per launch, {m['us_per_cycle']:.2f} per cycle")
