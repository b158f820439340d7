"""Sustained-clock check: per-chunk kernel time of the bench workload over many back-to-back steps."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from intfftk_amd import int_fft_single_path
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
chunk = 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
core = int_fft_single_path(NFFT=10, DATA_WIDTH=16, TWDL_WIDTH=16, FORMAT=0, RNDMODE=0)
x = bench.make_input(B, 0); y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps // chunk + 1)]
torch.cuda.synchronize()
evs[0].record()
for c in range(steps // chunk):
    for _ in range(chunk):
        core.exec_raw(x.data_ptr(), y.data_ptr(), B, st)
    evs[c + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) / chunk for i in range(len(evs) - 1)]
print("kernel=%s  us/step per %d-step chunk:" % (core.info["kernel_name"], chunk), " ".join("%.1f" % (m * 1e3) for m in ms))
print("batch=%d Gsample/s first=%.0f last=%.0f" % (B, B * 1024 / ms[0] / 1e6, B * 1024 / ms[-1] / 1e6))
