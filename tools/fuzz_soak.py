"""tools/fuzz_soak.py [seconds] [seed] -- open-ended parity fuzz on the GPU box (diagnostics; the seeded, bounded fuzz sets live in tests/).
Random elaboratable generics (NFFT 3..20, DATA_WIDTH 4..64, TWDL_WIDTH 8..26, every mode / direction / XSERIES / order pair, ragged
batches; now and then the 2-D scheme with a random split): whatever kernel the planner picks must equal the C oracle bit for bit.
FUZZ_BIG=1: the multi-pass families; FUZZ_LONG=1: the unscaled forward core at N = 2^17 .. 2^20; FUZZ_NATIVE=1: single cores in their own beat orders (HALVES / BITREV), widths weighted to the 32- / 64-bit word classes;
FUZZ_R6=1: the plans of round 6 -- BITREV_LANES at one end of a single core (store / load maps, BITREV twin + rotation) and USE_FLY = 0.
Prints one line per mismatch (none expected) and a summary of the kernels that were exercised."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from intfftk_amd import IntFFTCore
from oracle import oracle_c as C
from tests.helpers import edge_frames, uniform_frames

DIR = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
ORD = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
NP = {2: np.int16, 4: np.int32, 8: np.int64}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    seen = collections.Counter()
    bad = done = 0
    t0 = time.time()
    while time.time() - t0 < budget:
        big = os.environ.get("FUZZ_BIG") == "1"  # the multi-pass families: long frames, mostly 16-bit data, the tiled 2-D plans
        log2n = int(rng.choice([13, 14, 15, 16, 17, 18, 19, 19, 20, 20] if big else [3, 4, 5, 6, 6, 7, 7, 8, 9, 10, 10, 11, 12, 12, 13, 14, 15, 16, 17, 18, 19, 20]))
        fmt = int(rng.integers(0, 2))
        rnd = 0 if fmt else int(rng.integers(0, 2))
        dw = int(rng.choice([16, 16, 16, 16, 16, 12, 14, 24] if big else [16, 16, 16, 12, 14, 24, 32, int(rng.integers(4, 65))]))
        tw = int(rng.choice([16, 16, 24, int(rng.integers(8, 27))]))
        new = bool(rng.integers(0, 2))
        d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
        in_o, out_o = (list(ORD)[int(rng.integers(0, 4))], list(ORD)[int(rng.integers(0, 4))]) if rng.random() < 0.4 else ("NATURAL", "NATURAL")
        if os.environ.get("FUZZ_NATIVE") == "1":  # the cores' own beat orders (NAT instantiations): single cores, widths weighted to the 32- / 64-bit word classes
            d = ["FWD", "INV"][int(rng.integers(0, 2))]
            time_o, freq_o = [("HALVES", "BITREV"), ("HALVES", "NATURAL"), ("NATURAL", "BITREV")][int(rng.integers(0, 3))]
            in_o, out_o = (time_o, freq_o) if d == "FWD" else (freq_o, time_o)
            log2n = int(rng.choice([6, 7, 8, 9, 10, 10, 11, 11, 12, 12, 13, 14, 15, 16, 16, 17, 19, 20]))
            dw = int(rng.choice([16, 18, 24, 24, 28, 32, 32, 40, 48, int(rng.integers(4, 65))]))
        if os.environ.get("FUZZ_LONG") == "1":  # int_fftNk FORMAT = 1 at N = 2^17 .. 2^20 (csrc/intfft_widelong.hip): both width classes and their borders
            log2n, fmt, rnd, d, in_o, out_o = int(rng.integers(17, 21)), 1, 0, "FWD", "NATURAL", "NATURAL"
            dw = int(rng.choice([16, 16, 14, 15, 13, 20, 24, 24, 28, int(rng.integers(9, 33))]))
            tw = int(rng.choice([16, 16, 24, 12, int(rng.integers(8, 25))]))
        use_fly = 1
        if os.environ.get("FUZZ_R6") == "1":
            d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
            if rng.random() < 0.35:
                use_fly = 0
                in_o, out_o = list(ORD)[int(rng.integers(0, 4))], list(ORD)[int(rng.integers(0, 4))]
            else:
                d = ["FWD", "INV"][int(rng.integers(0, 2))]
                other = ["NATURAL", "HALVES", "BITREV"][int(rng.integers(0, 3))]
                in_o, out_o = (other, "BITREV_LANES") if d == "FWD" else ("BITREV_LANES", other)
            log2n = int(rng.choice([5, 6, 7, 8, 9, 10, 10, 11, 12, 12, 13, 14, 15, 16, 16, 17, 19, 20]))
            dw = int(rng.choice([16, 16, 16, 12, 18, 24, 24, 32, 40, int(rng.integers(4, 65))]))
        l1 = 0
        if big and log2n == 20 and rng.random() < 0.3:
            l1, log2n = 10, int(rng.choice([20, 20, 21, 21, 22, 22]))
            if log2n == 22 and rng.random() < 0.5:
                l1 = 11  # 2048 x 2048: the two-launch plans of round 5
        elif log2n >= 6 and rng.random() < 0.15:
            l1 = int(rng.integers(3, log2n - 2)) if rng.random() < 0.6 or log2n < 13 else 10 if log2n >= 20 else l1
        if use_fly == 0:
            l1 = 0
        p = C.make_params(log2n, dw, tw, fmt, rnd, new, use_fly)
        if l1:
            if in_o == "BITREV_LANES" or out_o == "BITREV_LANES" or C.lib().orc_validate_2d(p, l1, DIR[d]) != 0:
                continue
        elif C.lib().orc_validate(p, DIR[d]) != 0:
            continue
        n = 1 << log2n
        batch = int(rng.integers(1, 3)) if log2n >= 21 else int(rng.integers(1, 4)) if log2n >= 17 else int(rng.integers(1, 40)) if log2n >= 12 else int(rng.integers(1, 300))
        bits = dw if rng.random() < 0.5 else max(2, dw - 1)
        x = uniform_frames(batch, n, bits, int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.3:
            x = np.concatenate([x, edge_frames(n, dw)[: 1 + int(rng.integers(0, 6))]])
        try:
            core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", d, in_o, out_o, use_fly, NFFT1=l1)
        except Exception as exc:  # the planner refuses what the oracle accepts: report it
            print("PLAN-REFUSED", (log2n, dw, tw, fmt, rnd, new, d, in_o, out_o, l1), repr(exc)[:120], flush=True)
            bad += 1
            continue
        y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
        torch.cuda.synchronize()
        got = y.cpu().numpy().astype(np.int64)
        name = core.info["kernel_name"]
        core.close()
        want = (C.execute_2d(x, p, l1, DIR[d], ORD[in_o], ORD[out_o], form=1) if l1
                else C.execute(x, p, DIR[d], ORD[in_o], ORD[out_o], form=1))
        seen[name] += 1
        done += 1
        if got.shape != want.shape or not np.array_equal(got, want):
            bad += 1
            print("MISMATCH", (log2n, dw, tw, fmt, rnd, new, d, in_o, out_o, l1, batch, bits, use_fly), name, flush=True)
    print("fuzz_soak: %d configurations in %.0f s, %d mismatches" % (done, time.time() - t0, bad))
    for k, v in seen.most_common():
        print("  %5d  %s" % (v, k))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
