"""Throughput of a plan against the batch per call (is a MALL-sized working set faster per frame?)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from intfftk_amd import IntFFTCore

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
direction = sys.argv[2] if len(sys.argv) > 2 else "FWD"
core = IntFFTCore(log2n, 16, 16, 0, 0, "NEW", direction)
n = 1 << log2n
total = max(1, (1 << 30) // (4 * n))  # 1 GiB of input
x = torch.randint(-(1 << 14), 1 << 14, (total, n, 2), device="cuda", dtype=torch.int16)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for chunk in (total, 64, 32, 16, 8, 4, 2):
    if chunk > total:
        continue
    def run():
        for f0 in range(0, total, chunk):
            core.exec_raw(x[f0:].data_ptr(), y[f0:].data_ptr(), min(chunk, total - f0), st)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("N=2^%d %s frames/call=%d (%.0f MiB): %.1f Gsample/s" % (log2n, direction, chunk, chunk * n * 4 / 2**20, total * n / ms / 1e6), flush=True)
