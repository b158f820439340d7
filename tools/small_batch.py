"""Latency of small batches (one MI355X, data resident): us per call for nf frames, back-to-back calls on one stream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from intfftk_amd import IntFFTCore

CASES = [(L, 16, 16, 0, "FWD") for L in (5, 7, 10, 12, 13, 14, 16, 18, 20)] + [(10, 16, 16, 1, "FWD"), (13, 16, 16, 1, "FWD"),
                                                                               (16, 24, 24, 1, "FWD"), (12, 16, 16, 0, "PAIR")]
if __name__ == "__main__":
    for L, dw, tw, fmt, direction in CASES:
        core = IntFFTCore(L, dw, tw, fmt, 0, "NEW", direction, "NATURAL", "NATURAL")
        n = 1 << L
        row = []
        for nf in (1, 8, 64):
            if nf * n > (1 << 24):
                continue
            x = torch.randint(-1000, 1000, (nf, n, 2), device="cuda", dtype=core.in_dtype)
            y = torch.empty((nf, n, 2), device="cuda", dtype=core.out_dtype)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(200):
                core.exec_raw(x.data_ptr(), y.data_ptr(), nf, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(500):
                core.exec_raw(x.data_ptr(), y.data_ptr(), nf, st)
            e1.record()
            torch.cuda.synchronize()
            row.append("nf=%d: %.1f us" % (nf, e0.elapsed_time(e1) / 500 * 1000))
        print("N=2^%d dw=%d fmt=%d %s [%s]  " % (L, dw, fmt, direction, core.info["kernel_name"]) + ", ".join(row), flush=True)
        core.close()
