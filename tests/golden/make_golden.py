"""Regenerates tests/golden/*.npz and kats.json.

The reference (hukenovs/intfftk) holds NO golden vectors for this path and cannot be
run here (VHDL + Xilinx unisim; Octave absent), so these fixtures are produced by the build's own
independent pure-Python restatement of the RTL (oracle/oracle_py.py, literal bit-slicing style) and
pin BOTH the C oracle (tests/test_golden.py, CPU) and the HIP path (same file, -m gpu).
kats.json additionally records the survey-derived known-answer values of SURVEY.md section 8c, which were
obtained by a third, independent reading of the RTL.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as P  # noqa: E402
from tests.helpers import chirp_frame, edge_frames, to_list, uniform_frames  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {"TRUNCATE": (0, 0), "ROUNDING": (0, 1), "UNSCALED": (1, 0)}


def frames_for(n, dw, seed):
    e = edge_frames(n, dw)
    x = np.concatenate([uniform_frames(2, n, dw, seed), e[[1, 3, 4, 5]]])
    if n >= 128:
        x = np.concatenate([x, (chirp_frame(n) * (1 << (dw - 10)))[None]])
    return x


def sample_index(s):
    n = 1 << s
    return np.unique(np.concatenate([np.arange(64), np.arange(0, n, 1021), np.arange(n - 64, n),
                                     np.arange(n // 2 - 32, n // 2 + 32)])).astype(np.int64)


def main():
    cases = {}
    for log2n in (3, 4, 7, 10):
        n = 1 << log2n
        for mode, (fmt, rnd) in MODES.items():
            for direction, dname in ((P.FWD, "FWD"), (P.INV, "INV"), (P.PAIR, "PAIR")):
                if log2n == 10 and direction != P.FWD and mode == "ROUNDING":
                    continue  # keep the fixture small
                x = frames_for(n, 16, 100 + log2n)
                if log2n == 10:
                    x = x[[0, 3, 4, 6]]
                y = np.array([P.execute(to_list(f), log2n, 16, 16, fmt, rnd, True, direction) for f in x],
                             dtype=np.int64)
                key = "n%d_%s_%s_w16_t16_NEW" % (n, mode, dname)
                cases[key + "_in"] = x.astype(np.int64)
                cases[key + "_out"] = y
    # wide-multiplier regimes and XSER dependence, small N
    for (log2n, dw, tw, fmt, new) in [(4, 24, 24, 1, True), (4, 30, 16, 1, True), (4, 30, 16, 1, False),
                                      (4, 44, 16, 1, True), (5, 32, 24, 1, True), (4, 14, 24, 1, True),
                                      (4, 16, 18, 0, True),
                                      # trpl18 beyond its A port (SXT(M_AA, 61 | 59) cuts the operand, int_cmult_trpl18_dsp48.vhd:161-162),
                                      # at the widest multiplier that elaborates with 16-bit twiddles (MAW + MBW = 80 | 78, :151-152)
                                      (4, 64, 16, 0, True), (4, 62, 16, 0, False)]:
        x = frames_for(1 << log2n, dw, 7)
        for direction, dname in ((P.FWD, "FWD"), (P.INV, "INV")):
            y = np.array([P.execute(to_list(f), log2n, dw, tw, fmt, 0, new, direction) for f in x], dtype=np.int64)
            key = "n%d_%s_%s_w%d_t%d_%s" % (1 << log2n, "UNSCALED" if fmt else "TRUNCATE", dname, dw, tw,
                                            "NEW" if new else "OLD")
            cases[key + "_in"] = x.astype(np.int64)
            cases[key + "_out"] = y
    # narrow data in int16 containers (DATA_WIDTH 9 .. 15: 12- / 14-bit converters), both scaled modes: w-bit product slices, the
    # w-bit wrap of the rhu2 difference (+2^(w-1) -> -2^(w-1): the "hi / lo" frames reach it), containers that hold more than w bits
    for (log2n, dw, tw, mode) in [(7, 12, 16, "TRUNCATE"), (7, 12, 16, "ROUNDING"), (7, 14, 12, "ROUNDING"), (5, 9, 16, "ROUNDING"),
                                  (10, 14, 16, "ROUNDING"), (10, 12, 16, "TRUNCATE")]:
        n = 1 << log2n
        fmt, rnd = MODES[mode]
        x = frames_for(n, dw, 31 + dw)
        hi, lo = (1 << (dw - 1)) - 1, -(1 << (dw - 1))
        alt = np.empty((2, n, 2), dtype=np.int64)
        alt[0, 0::2], alt[0, 1::2] = hi, lo
        alt[1, : n // 2], alt[1, n // 2 :] = hi, lo
        x = np.concatenate([x[:3] if log2n == 10 else x, alt, uniform_frames(1, n, 16, 77 + dw)])
        for direction, dname in ((P.FWD, "FWD"), (P.INV, "INV"), (P.PAIR, "PAIR")):
            if log2n == 10 and direction != P.FWD:
                continue
            y = np.array([P.execute(to_list(f), log2n, dw, tw, fmt, rnd, True, direction) for f in x], dtype=np.int64)
            key = "n%d_%s_%s_w%d_t%d_NEW" % (n, mode, dname, dw, tw)
            cases[key + "_in"] = x.astype(np.int64)
            cases[key + "_out"] = y
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **cases)

    # twiddle streams: full tables up to STAGE 12; beyond that a strided sample + CRC32 of the table
    import zlib

    tw = {}
    for t in (16, 24):
        for new in (True, False):
            for s in range(20):
                full = np.array(P.twiddles(s, t, new), dtype=np.int32)
                key = "s%d_t%d_%s" % (s, t, "NEW" if new else "OLD")
                if s <= 12:
                    tw[key] = full
                else:
                    idx = sample_index(s)
                    tw[key + "_idx"] = idx
                    tw[key + "_val"] = full[idx]
                    tw[key + "_crc"] = np.array([zlib.crc32(full.tobytes())], dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "twiddles.npz"), **tw)

    kats = {
        "source": "SURVEY.md section 8c 'Survey-derived KATs' (independent scratch reading of the RTL)",
        "twiddles": [
            {"stage": 3, "t": 16, "new": True, "first": [[32767, 0], [30273, -12539], [23170, -23170], [12539, -30273],
                                                          [0, -32767], [-12539, -30273], [-23170, -23170], [-30273, -12539]]},
            {"stage": 2, "t": 24, "new": True, "first": [[4194303, 0], [2965820, -2965820], [0, -4194303], [-2965820, -2965820]]},
            {"stage": 11, "t": 16, "new": True, "first": [[32767, 0], [32767, -50], [32767, -101]]},
            {"stage": 15, "t": 24, "new": True, "first": [[4194303, 0], [4194303, -402], [4194303, -804]]},
        ],
        "cmult": [
            [-12345, 23456, 23170, -23170, 16, 16, True, "sngl", 7856, 25314],
            [32767, -32768, -30273, -12539, 16, 16, True, "sngl", 22724, 17734],
            [-123456789, 98765432, 30273, -12539, 30, 16, True, "dbl18", -76263050, 138487261],
            [-123456789, 98765432, 30273, -12539, 30, 16, False, "dbl18", -76263050, 138487261],
            [291770562, 216703618, 21856, -24413, 30, 16, True, "dbl18", 356058436, -72836929],
            [291770562, 216703618, 21856, -24413, 30, 16, False, "dbl18", 356058435, -72836928],
            [-123456789012, 98765432101, 12539, -30273, 46, 16, True, "trpl18", 44003334002, 151850193080],
            [-30000, 29999, 2965820, -2965820, 17, 24, True, "sngl25", -1, 42425],
            [-123456789, 98765432, 2965820, -2965820, 30, 24, True, "dbl35", -17459422, 157134796],
            [-123456789012, 98765432101, -2965820, -2965820, 40, 24, True, "trpl52", 157134797054, 17459421194],
            # trpl18 beyond its A port (round 2, from reading int_cmult_trpl18_dsp48.vhd:161-162: dspA <= SXT(M_AA, 61 | 59) keeps the
            # low 61 | 59 bits): 2^62 has none of them set -> the product is 0; 2^60 + 5 reads as -2^60 + 5 and -2^61 as 0;
            # OLD: 2^58 + 7 reads as -2^58 + 7; a 61-bit operand passes unchanged
            [4611686018427387904, 0, 32767, 0, 64, 16, True, "trpl18", 0, 0],
            [1152921504606846981, -2305843009213693952, 30273, -12539, 64, 16, True, "trpl18", -1065136496245211132, 441176841621864446],
            [288230376151711751, 3, 23170, -23170, 62, 16, False, "trpl18", -203805475324559353, 203805475324559357],
            [576460752303423487, -576460752303423488, 32767, 0, 61, 16, True, "trpl18", 576443160117379071, -576443160117379072],
        ],
        "frames": [
            {"log2n": 3, "dw": 16, "tw": 16, "fmt": 0, "rnd": 0, "dir": "FWD",
             "in": [[(k + 1) * 1000, -(k + 1) * 500] for k in range(8)],
             "out": [[4500, -2250], [102, 1455], [-250, 749], [-397, 456], [-500, 250], [-604, 43], [-750, -249], [-1103, -956]]},
            {"log2n": 3, "dw": 16, "tw": 16, "fmt": 0, "rnd": 1, "dir": "FWD",
             "in": [[(k + 1) * 1000, -(k + 1) * 500] for k in range(8)],
             "out": [[4500, -2250], [104, 1457], [-250, 750], [-396, 457], [-500, 250], [-603, 43], [-750, -249], [-1103, -956]]},
            {"log2n": 4, "dw": 16, "tw": 16, "fmt": 1, "rnd": 0, "dir": "FWD",
             "in": [[1000 if k == 1 else 0, 0] for k in range(16)],
             "out": [[1000, 0], [923, -383], [707, -708], [381, -924], [0, -1000], [-383, -923], [-708, -707], [-924, -381],
                     [-1000, 0], [-923, 383], [-707, 708], [-381, 924], [0, 1000], [383, 923], [708, 707], [924, 381]]},
            {"log2n": 4, "dw": 16, "tw": 16, "fmt": 1, "rnd": 0, "dir": "FWD",
             "in": [[-1000 if k == 1 else 0, 0] for k in range(16)],
             "out": [[-1000, 0], [-924, 382], [-708, 707], [-384, 923], [0, 999], [382, 923], [707, 707], [923, 383],
                     [1000, 0], [924, -382], [708, -707], [384, -923], [0, -999], [-382, -923], [-707, -707], [-923, -383]]},
            {"log2n": 4, "dw": 20, "tw": 16, "fmt": 1, "rnd": 0, "dir": "INV", "in_order": "BITREV",
             "in": [[1000, 0], [-1000, 0], [0, -1000], [0, 1000], [707, -708], [-707, 708], [-708, -707], [708, 707],
                    [923, -383], [-923, 383], [-383, -923], [383, 923], [381, -924], [-381, 924], [-924, -381], [924, 381]],
             "out": [[0, 0], [15993, -15], [0, 0], [0, 0], [0, 0], [-8, 3], [0, 0], [0, 0], [0, 0], [7, 7], [0, 0], [0, 0],
                     [0, 0], [4, 5], [0, 0], [4, 0]]},
            {"log2n": 3, "dw": 16, "tw": 16, "fmt": 0, "rnd": 0, "dir": "PAIR",
             "in": [[(k + 1) * 1000, -(k + 1) * 500] for k in range(8)],
             "out": [[124, -64], [250, -126], [375, -188], [500, -251], [626, -312], [750, -374], [875, -436], [1000, -499]]},
        ],
    }
    with open(os.path.join(HERE, "kats.json"), "w") as fh:
        json.dump(kats, fh, indent=1)
    print("wrote", len(cases) // 2, "frame cases,", len(tw), "twiddle tables, kats.json")


if __name__ == "__main__":
    main()
