"""Stress: interleave intfft_exec_host (hipHostRegister on numpy memory) with ordinary pageable copies of many sizes,
looking for stale-registration faults.  argv[1] = 'host' to include exec_host calls, 'nohost' to leave them out."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from intfftk_amd import IntFFTCore

mode = sys.argv[1] if len(sys.argv) > 1 else "host"
rng = np.random.default_rng(1)
core = IntFFTCore(10, 16, 16, 0, 0, "NEW", "FWD")
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    if mode == "host":
        b = int(rng.integers(1, 80))
        x = rng.integers(-1000, 1000, size=(b, 1024, 2)).astype(np.int16)
        y = core.exec_host(x, int(rng.integers(0, 9)))
        del x, y
    for _ in range(20):
        b = int(rng.integers(1, 300))
        x = rng.integers(-1000, 1000, size=(b, 1024, 2)).astype(np.int16)
        xd = torch.from_numpy(x).cuda()
        yd = core(xd)
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        del x, xd, yd, y
    if it % 50 == 0:
        print("iter", it, flush=True)
print("stress ok", mode)
