"""What the device sustains for PURE WRITES (the rollout kernels write 50 B and read ~2 B per env-step) next to a copy:
torch fill_ / copy_ on 2 GiB tensors at steady clocks."""
import time
import torch
dev = 'cuda:0'
n = 512 * 1024 * 1024                      # 2 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:       # settle the clock
    x.fill_(1.0); torch.cuda.synchronize()
def rate(fn, bytes_, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return bytes_ * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
print('fill_  (write only)      %.0f GB/s' % rate(lambda: x.fill_(2.0), 4 * n))
print('zero_  (write only)      %.0f GB/s' % rate(lambda: x.zero_(), 4 * n))
print('copy_  (read + write)    %.0f GB/s total' % rate(lambda: y.copy_(x), 8 * n))
print('sum    (read only)       %.0f GB/s' % rate(lambda: x.sum(), 4 * n))
