import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from intfftk_amd import IntFFTCore
for L, dw, fmt in [(13, 16, 0), (14, 16, 0), (13, 16, 1), (14, 16, 1)]:
    core = IntFFTCore(L, dw, 16, fmt, 0, "NEW", "FWD", "NATURAL", "NATURAL")
    n = 1 << L
    for nf in (1, 4, 16, 32, 64, 128, 256, 512):
        x = torch.randint(-1000, 1000, (nf, n, 2), device="cuda", dtype=core.in_dtype)
        y = torch.empty((nf, n, 2), device="cuda", dtype=core.out_dtype)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(200): core.exec_raw(x.data_ptr(), y.data_ptr(), nf, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500): core.exec_raw(x.data_ptr(), y.data_ptr(), nf, st)
        e1.record(); torch.cuda.synchronize()
        print(L, fmt, nf, "log2 samples", L + (nf.bit_length() - 1), "us", round(e0.elapsed_time(e1) / 500 * 1000, 2), flush=True)
    core.close()
