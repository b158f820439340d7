"""tools/fuzz_wide.py [seconds] [seed] -- parity fuzz for results beyond 64 bits (k_pass<__int128>, 16-byte containers) and for the
64-bit boundary, checked against the pure-Python twin (big integers; the C oracle stops at 64 bits).  Small frames: the twin is slow."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from intfftk_amd import IntFFTCore
from intfftk_amd import _capi as capi
from intfftk_amd.engine import wide_to_int
from oracle import oracle_py as P
from tests.helpers import edge_frames, uniform_frames

NP = {2: np.int16, 4: np.int32, 8: np.int64}
DIR = {"FWD": P.FWD, "INV": P.INV, "PAIR": P.PAIR}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
    seen = collections.Counter()
    bad = done = refused = 0
    t0 = time.time()
    while time.time() - t0 < budget:
        log2n = int(rng.integers(3, 9))
        dw = int(rng.integers(30, 65))
        tw = int(rng.integers(8, 27))
        new = bool(rng.integers(0, 2))
        d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
        fmt = 1 if rng.random() < 0.8 else 0
        rnd = 0 if fmt else int(rng.integers(0, 2))
        ob = dw + fmt * log2n * (2 if d == "PAIR" else 1)
        if ob < 60 or ob > 96:
            continue
        try:
            core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", d)
        except Exception:
            refused += 1  # not elaboratable (the planner's accept set is the oracle's: tests/test_capi_cpu.py)
            continue
        n = 1 << log2n
        x = np.concatenate([uniform_frames(int(rng.integers(1, 4)), n, dw, int(rng.integers(1, 1 << 30))), edge_frames(n, dw)[[1, 4]]])
        y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
        torch.cuda.synchronize()
        info = dict(core.info)
        core.close()
        y = y.cpu().numpy()
        got = wide_to_int(y) if info["out_container"] == 16 else y.astype(object)
        ok = True
        for f in range(x.shape[0]):
            want = P.execute([(int(a), int(b)) for a, b in x[f]], log2n, dw, tw, fmt, rnd, new, DIR[d])
            for m, (wr, wi) in enumerate(want):
                if (int(got[f, m, 0]), int(got[f, m, 1])) != (wr, wi):
                    ok = False
                    break
            if not ok:
                break
        seen[info["kernel_name"]] += 1
        done += 1
        if not ok:
            bad += 1
            print("MISMATCH", (log2n, dw, tw, fmt, rnd, new, d), info["kernel_name"], "out_bits", info["out_bits"], flush=True)
    print("fuzz_wide: %d configurations (%d refused) in %.0f s, %d mismatches" % (done, refused, time.time() - t0, bad))
    for k, v in seen.most_common():
        print("  %5d  %s" % (v, k))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
