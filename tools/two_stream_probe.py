"""Does running two chunks of a two-pass plan on two streams (pass A of one beside pass B of the other) beat one stream?
(diagnostic probe for DESIGN.md section 4.2d; INTFFT_SCRATCH_MB bounds each plan's scratch)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("INTFFT_DIAG", "1")
import torch

from intfftk_amd import IntFFTCore

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = 1 << log2n
x = torch.randint(-(1 << 14), 1 << 14, (batch, n, 2), device="cuda", dtype=torch.int16)
y = torch.empty_like(x)
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cores = [IntFFTCore(log2n, 16, 16, 0, 0, "NEW", "FWD", "NATURAL", "NATURAL") for _ in range(ns)]
s = [torch.cuda.Stream() for _ in range(ns)]
half = batch // ns


def one():
    cores[0].exec_raw(x.data_ptr(), y.data_ptr(), batch, s[0].cuda_stream)


def two(offset_frames=0):
    fb = 2 * n * 2
    for i in range(ns):
        cnt = half if i < ns - 1 else batch - half * (ns - 1)
        cores[i].exec_raw(x.data_ptr() + i * half * fb, y.data_ptr() + i * half * fb, cnt, s[i].cuda_stream)


for name, fn in (("one stream", one), ("%d streams" % ns, two), ("one stream", one), ("%d streams" % ns, two)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%-12s %.3f ms  %.1f Gsample/s" % (name, dt * 1e3, batch * n / dt / 1e9), flush=True)
