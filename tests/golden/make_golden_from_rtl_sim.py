"""Regenerates tests/golden/from_rtl_sim.npz: WHOLE FRAMES computed by clocking the reference's own VHDL text.

tools/rtl_sim.py elaborates int_fftNk / int_ifftNk and the two wrappers of src/vhdl/main (I/O buffers included) from the reference's files,
DSP48 pipeline registers as their generic maps ask, and runs frames through them clock by clock.  This script stores what went in and
what came out -- data, no text -- so that the comparison also runs where the reference does not exist (the GPU box):

  <case>_x    int64 (frames, N, 2)   the frames fed in, in the memory order named by the case's `in` order
  <case>_y    int64 (frames', N, 2)  the frames that came out with the valid strobe, memory order of the case's `out` order
                                     (frames' < frames for the wrappers: their buffers hand a frame out while the next comes in)
  cases       the table: name, entity, direction, NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, in order, out order

  core_fwd_*  int_fftNk:  natural frames in (lane 0 = x[i], lane 1 = x[i + N/2]), the core's bit-reversed pair stream out = BITREV memory
  core_inv_*  int_ifftNk: the bit-reversed pair stream in = BITREV memory, natural frames out
  single_*    int_fft_single_path: one sample per clock, natural in, natural out
  pair_*      int_fft_ifft_pair: two samples per clock (x[2i], x[2i + 1]) = natural memory in and out, direction PAIR

Nothing here comes from the oracle; the one model that is not the reference's text is the DSP48 slice (oracle/dsp48_twin.dsp48, UG479 /
UG579).  tests/test_golden_from_rtl_sim.py holds the C oracle and the Python twin (CPU) and the HIP path through the C-ABI (-m gpu)
against the file.

Run from the repo root, in the container that has /root/reference:  python tests/golden/make_golden_from_rtl_sim.py     (~40 min)
"""
import os
import random
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import rtl_sim as S  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (entity kind, NFFT, DATA_WIDTH, TWDL_WIDTH, FORMAT, RNDMODE, XSER, frames)
CASES = (
    [("core_fwd", n, 16, 16, f, r, x, 3) for n in (3, 4, 5) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "NEW"), (1, 0, "NEW"), (0, 0, "OLD"))]
    + [("core_inv", n, 16, 16, f, r, x, 3) for n in (3, 4, 5) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "OLD"), (1, 0, "NEW"))]
    + [("core_fwd", 3, 24, 24, 1, 0, "NEW", 3), ("core_inv", 3, 24, 24, 0, 1, "OLD", 3), ("core_fwd", 3, 30, 16, 1, 0, "OLD", 3),
       ("core_inv", 3, 32, 16, 0, 0, "NEW", 3), ("core_fwd", 4, 40, 24, 1, 0, "NEW", 2), ("core_fwd", 3, 12, 10, 0, 0, "NEW", 3),
       ("core_fwd", 6, 16, 16, 0, 0, "NEW", 2), ("core_inv", 6, 16, 16, 0, 0, "NEW", 2)]
    + [("single", n, 16, 16, f, r, x, 4) for n in (3, 4, 5) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "OLD"), (1, 0, "NEW"))]
    + [("single", 3, 24, 24, 1, 0, "OLD", 4), ("single", 6, 16, 16, 0, 0, "NEW", 3)]
    + [("pair", n, 16, 16, f, r, x, 4) for n in (3, 4) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "NEW"), (1, 0, "OLD"))]
    + [("pair", 5, 16, 16, 0, 0, "NEW", 4)]
    # the shapes of BASELINE.json at the sizes a Python-clocked core still reaches: C2 (N = 1024, 16 / 16 scaled, natural in and out through
    # the single-path wrapper; the core alone in the three modes and both series), C5 (the N = 4096 pair), and N = 4096 alone, where STAGE 11
    # takes its twiddles from row_twiddle_tay
    + [("core_fwd", 10, 16, 16, f, r, x, 2) for (f, r, x) in ((0, 0, "NEW"), (0, 1, "NEW"), (1, 0, "NEW"), (0, 0, "OLD"))]
    + [("core_inv", 10, 16, 16, 0, 0, "NEW", 2), ("core_inv", 10, 16, 16, 0, 1, "OLD", 2), ("single", 10, 16, 16, 0, 0, "NEW", 3),
       ("core_fwd", 12, 16, 16, 0, 0, "NEW", 2), ("core_inv", 12, 16, 16, 0, 0, "NEW", 2), ("core_fwd", 12, 16, 16, 0, 1, "OLD", 1),
       ("core_fwd", 12, 24, 24, 1, 0, "NEW", 1), ("core_fwd", 13, 16, 16, 0, 0, "NEW", 1), ("core_fwd", 8, 18, 16, 0, 0, "NEW", 2),
       ("core_inv", 9, 20, 18, 1, 0, "OLD", 2), ("pair", 7, 16, 16, 0, 0, "NEW", 4), ("pair", 12, 16, 16, 0, 0, "NEW", 3)])


def main():
    if not S.available():
        sys.exit("the reference's text is not on this machine")
    out, table = {}, []
    for (kind, nfft, dw, t, f, r, x, count) in CASES:
        name = "%s_n%d_w%d_t%d_f%d_r%d_%s" % (kind, nfft, dw, t, f, r, x)
        n = 1 << nfft
        frames = S._frames(random.Random(zlib.crc32(name.encode())), nfft, dw, count)
        if kind in ("core_fwd", "core_inv"):
            d = "FWD" if kind == "core_fwd" else "INV"
            beats, _ = S.run_core(d, nfft, dw, t, f, r, x, frames, "cont")
            assert len(beats) == count * n // 2, (name, len(beats))
            y = []
            for k in range(count):
                fb = beats[k * n // 2:(k + 1) * n // 2]
                # forward: beat i = (v[2i], v[2i + 1]) of the bit-reversed sequence; inverse: beat i = (y[i], y[i + N/2])
                y.append([s for b in fb for s in b] if d == "FWD" else [b[0] for b in fb] + [b[1] for b in fb])
            orders = ("NATURAL", "BITREV") if d == "FWD" else ("BITREV", "NATURAL")
        elif kind == "single":
            d = "FWD"
            got, _ = S.run_single_path(nfft, dw, t, f, r, x, frames, flush=1 if nfft >= 7 else 0)
            y = [got[k * n:(k + 1) * n] for k in range(min(count, len(got) // n))]
            orders = ("NATURAL", "NATURAL")
        else:
            d = "PAIR"
            beats, _ = S.run_pair(nfft, dw, t, f, r, x, frames, flush=2 if nfft >= 7 else 0)   # (two all-zero frames push the last ones out)
            got = [s for b in beats for s in b]
            y = [got[k * n:(k + 1) * n] for k in range(min(count, len(got) // n))]
            orders = ("NATURAL", "NATURAL")
        assert len(y) >= 1, name
        out[name + "_x"] = np.array(frames, dtype=np.int64)
        out[name + "_y"] = np.array(y, dtype=np.int64)
        table.append("%s %s %s %d %d %d %d %d %s %s %s" % (name, kind, d, nfft, dw, t, f, r, x, orders[0], orders[1]))
        print(table[-1], "->", len(y), "of", count, "frames", flush=True)
    out["cases"] = np.array(table)
    np.savez_compressed(os.path.join(HERE, "from_rtl_sim.npz"), **out)
    print("wrote from_rtl_sim.npz:", len(table), "cases")


def big(beats_from=None):
    """--big: ONE full-size frame of BASELINE's C3 shape (N = 65536, 24-bit data, 24-bit twiddles, unscaled: 40-bit results; STAGE 11 .. 15
    on Taylor twiddles, the sngl25 -> dbl35 -> trpl52 walk of the multipliers) through int_fftNk from the text -> from_rtl_sim_big.npz.
    About 100 000 simulated clocks of a sixteen-stage core: hours of CPU."""
    import pickle
    nfft, dw, t, f, r, x = 16, 24, 24, 1, 0, "NEW"
    frames = S._frames(random.Random(65536), nfft, dw, 1)
    if beats_from:   # a run of exactly this call that was left to finish on its own (same seed, same generics), stored as (frames, beats)
        fr2, beats = pickle.load(open(beats_from, "rb"))
        assert fr2 == frames
    else:
        beats, _ = S.run_core("FWD", nfft, dw, t, f, r, x, frames, "cont")
    n = 1 << nfft
    assert len(beats) >= n // 2
    y = [[s for b in beats[:n // 2] for s in b]]
    name = "core_fwd_n16_w24_t24_f1_r0_NEW"
    np.savez_compressed(os.path.join(HERE, "from_rtl_sim_big.npz"), **{name + "_x": np.array(frames, dtype=np.int64), name + "_y": np.array(y, dtype=np.int64),
                        "cases": np.array(["%s core_fwd FWD 16 24 24 1 0 NEW NATURAL BITREV" % name])})
    print("wrote from_rtl_sim_big.npz")


if __name__ == "__main__":
    if "--big" in sys.argv:
        big(sys.argv[sys.argv.index("--from") + 1] if "--from" in sys.argv else None)
    else:
        main()
