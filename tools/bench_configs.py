"""Secondary measurements (NOT the bench.py contract line): every BASELINE config on one GPU.
Gsample/s, algorithmic GB/s, roofline fraction, and a parity check of a frame prefix against the oracle."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("INTFFT_DIAG", "1")  # a diagnostics tool: INTFFT_NO_* A/B switches are honoured (bench.py never sets this)
import numpy as np
import torch

from intfftk_amd import IntFFTCore
from oracle import oracle_c as C

CONFIGS = {
    # name: (log2n, dw, tw, fmt, rnd, direction, batch, bits of the random data, bytes/sample)
    "C2": (10, 16, 16, 0, 0, "FWD", 65536, 15, 8),
    "C2r": (10, 16, 16, 0, 1, "FWD", 65536, 15, 8),
    "C3": (16, 24, 24, 1, 0, "FWD", 4096, 23, 24),
    "C3t16": (16, 24, 16, 1, 0, "FWD", 4096, 23, 24),
    "C4": (20, 16, 16, 0, 0, "FWD", 1024, 15, 8),
    "C5": (12, 16, 16, 0, 0, "PAIR", 16384, 15, 8),
    "C2inv": (10, 16, 16, 0, 0, "INV", 65536, 15, 8),
    "C2u": (10, 16, 16, 1, 0, "FWD", 65536, 15, 12),
    "tb7u": (7, 16, 16, 1, 0, "FWD", 524288, 15, 12),
    "C2pair": (10, 16, 16, 0, 0, "PAIR", 65536, 15, 8),
    "C5fwd": (12, 16, 16, 0, 0, "FWD", 16384, 15, 8),
    "C5inv": (12, 16, 16, 0, 0, "INV", 16384, 15, 8),
}


ORDERS = {"C2native": ("HALVES", "BITREV")}
USE_FLY = {}  # spec -> 0 for the bypass mux (10th field of an ad-hoc spec)
NFFT1 = {}  # spec -> log2 N1 of a 2-D scheme plan (spec "L:DW:TW:FMT:RND:DIR:L1")
CONFIGS["C2native"] = (10, 16, 16, 0, 0, "FWD", 65536, 15, 8)


def run(name, steps=None, check_frames=8):
    steps = steps or int(os.environ.get("BENCH_STEPS", "20"))  # (PMC passes serialise every launch: BENCH_STEPS / BENCH_RAMP_S shorten them)
    log2n, dw, tw, fmt, rnd, direction, batch, bits, bps = CONFIGS[name]
    n = 1 << log2n
    in_o, out_o = ORDERS.get(name, ("NATURAL", "NATURAL"))
    l1 = NFFT1.get(name, 0)
    uf = USE_FLY.get(name, 1)
    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW", direction, in_o, out_o, uf, NFFT1=l1)
    g = torch.Generator(device="cuda")
    g.manual_seed(0xC0FFEE00 + log2n)
    x = torch.randint(-(1 << (bits - 1)), 1 << (bits - 1), (batch, n, 2), device="cuda", dtype=core.in_dtype, generator=g)
    y = torch.empty(core.out_shape(batch), device="cuda", dtype=core.out_dtype)
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    calls = steps
    while time.time() - t0 < float(os.environ.get("BENCH_RAMP_S", "0.25")):  # clock ramp
        for _ in range(10):
            core.exec_raw(x.data_ptr(), y.data_ptr(), batch, st)
        calls += 10
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        core.exec_raw(x.data_ptr(), y.data_ptr(), batch, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    p = C.make_params(log2n, dw, tw, fmt, rnd, True, uf)
    om = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
    dd = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}[direction]
    if core.out_container == 16:  # results beyond 64 bits: the Python twin, small lengths only (tests/test_gpu_wide128.py)
        from intfftk_amd.engine import wide_to_int
        from oracle import oracle_py as PY

        check_frames = min(check_frames, 1) if n <= 4096 else 0
        got = wide_to_int(y[:check_frames].cpu().numpy())
        ok = True
        for f in range(check_frames):
            w = PY.execute([(int(a), int(b)) for a, b in x[f].cpu().numpy()], log2n, dw, tw, fmt, rnd, True, dd, om[in_o], om[out_o])
            ok = ok and all((got[f, m, 0], got[f, m, 1]) == w[m] for m in range(n))
        want = None
    elif l1:
        check_frames = min(check_frames, 2)
        want = C.execute_2d(x[:check_frames].cpu().numpy(), p, l1, dd, om[in_o], om[out_o])
    else:
        want = C.execute(x[:check_frames].cpu().numpy(), p, dd, om[in_o], om[out_o])
    if want is not None:
        ok = bool(np.array_equal(y[:check_frames].cpu().numpy().astype(np.int64), want))
    gs = batch * n / ms / 1e6
    out = {"config": name, "log2n": log2n, "batch": batch, "dir": direction, "ms": ms, "Gsample/s": gs,
           "GB/s": gs * bps, "roofline_frac": gs * bps / 8000.0, "passes": core.info["n_passes"],
           "kernel": core.info["kernel_name"], "parity_prefix_ok": ok, "calls": calls}
    core.close()
    del x, y
    torch.cuda.empty_cache()
    return out


def adhoc(spec):
    """spec "L:DW:TW:FMT[:RND[:DIR[:L1[:IN_ORDER:OUT_ORDER]]]]" -> a 256 MiB-input config with that shape, e.g. 11:16:16:0 (L1: 2-D scheme,
    N1 = 2^L1, 0 = none), 10:16:16:0:0:FWD:0:NATURAL:BITREV_LANES"""
    f = spec.split(":")
    log2n, dw, tw, fmt = (int(v) for v in f[:4])
    rnd = int(f[4]) if len(f) > 4 else 0
    direction = f[5] if len(f) > 5 else "FWD"
    if len(f) > 6 and int(f[6]):
        NFFT1[spec] = int(f[6])
    if len(f) > 8:
        ORDERS[spec] = (f[7], f[8])
    if len(f) > 9:
        USE_FLY[spec] = int(f[9])
    in_cb = 2 if dw <= 16 else 4 if dw <= 32 else 8
    ob = dw + (fmt * log2n) * (2 if direction == "PAIR" else 1)
    out_cb = 2 if ob <= 16 else 4 if ob <= 32 else 8 if ob <= 64 else 16
    batch = max(1, (int(os.environ.get("BENCH_INPUT_MB", "256")) << 20) // ((2 * in_cb) << log2n))
    CONFIGS[spec] = (log2n, dw, tw, fmt, rnd, direction, batch, min(dw, 31) - 1, 2 * (in_cb + out_cb))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CONFIGS)
    for nm in names:
        if nm not in CONFIGS:
            adhoc(nm)
        print(json.dumps(run(nm)), flush=True)
