"""Regenerates tests/golden/from_rtl_text.npz: vectors computed from the REFERENCE'S OWN VHDL TEXT.

tools/rtl_interp.py parses and evaluates the reference's files where the reference tree exists (this container).  This script runs it and
stores only inputs and outputs -- data, no text:

  tw_s<STAGE>_t<AWD>_<XSER>          full twiddle tables of STAGE 2 .. 10 (rom_twiddle_int: the ROM its function fills from MATH_PI, COS, SIN
                                     and the quadrant rotation; no DSP48 is involved below STAGE 11, so these come from the text alone)
  tw_s<STAGE>_t<AWD>_<XSER>_idx/val  STAGE 11 .. 18 at sampled counter values (the Taylor correction adds two DSP48 slices: text + the slice
                                     model of oracle/dsp48_twin.py)
  cm_<w>_<t>_<XSER>_in / _out        int_cmult_dsp48 on random and corner operands, every regime (text + slice model)
  fly_<kind>_<...>_in / _out         int_dif2_fly / int_dit2_fly, the three modes, STAGE 0 / 1 (both toggles) / n (text + slice model)

The fixture travels to the GPU box, where the reference does not exist: tests/test_golden_from_rtl.py holds the oracle (CPU) and the HIP
path (-m gpu: intfft_twiddles of real plans, one-butterfly-deep checks are the oracle's job) against it.

Run from the repo root, in the container that has /root/reference:  python tests/golden/make_golden_from_rtl.py
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import rtl_interp as R  # noqa: E402
from oracle import dsp48_twin as tw  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CMULT = [(16, 16), (12, 10), (27, 16), (25, 16), (30, 16), (40, 12), (44, 16), (46, 16), (60, 16), (17, 24), (18, 19), (24, 24), (30, 24),
         (35, 27), (40, 24), (52, 24)]
FLY = [  # (dtw, tfw, scale, rndmode, stage, odd)
    (16, 16, 1, 0, 0, 0), (16, 16, 1, 0, 1, 0), (16, 16, 1, 0, 1, 1), (16, 16, 1, 0, 7, 0), (16, 16, 1, 1, 0, 0), (16, 16, 1, 1, 1, 1),
    (16, 16, 1, 1, 5, 0), (16, 16, 0, 0, 1, 1), (16, 16, 0, 0, 9, 0), (24, 24, 0, 0, 12, 0), (30, 16, 0, 0, 3, 0), (12, 16, 1, 0, 4, 0),
]


def twiddle_value(stage, awd, xser, cnt):
    r = R.evaluate("rom_twiddle_int", {"awd": awd, "nfft": 20, "stage": stage, "use_mlt": False, "xser": xser.lower()},
                   {"cnt": cnt, "rst": 0, "ww_en": 1})
    return tw.signed(r["ww_re"], awd), tw.signed(r["ww_im"], awd)


def main():
    assert R.available(), "the reference tree is not present: this script only runs where it is"
    rng = random.Random(20260930)
    out = {}
    for xser in ("NEW", "OLD"):
        for awd in (16, 24):
            for stage in range(2, 11):
                out["tw_s%d_t%d_%s" % (stage, awd, xser)] = np.array([twiddle_value(stage, awd, xser, c) for c in range(1 << stage)], dtype=np.int32)
            for stage in range(11, 19):
                idx = sorted(set(list(range(40)) + [rng.randrange(1 << stage) for _ in range(160)] + [(1 << stage) - 1 - k for k in range(24)]
                                 + [(1 << (stage - 1)) + k for k in range(-12, 12)]))
                out["tw_s%d_t%d_%s_idx" % (stage, awd, xser)] = np.array(idx, dtype=np.int64)
                out["tw_s%d_t%d_%s_val" % (stage, awd, xser)] = np.array([twiddle_value(stage, awd, xser, c) for c in idx], dtype=np.int32)
        for w, t in CMULT:
            from oracle import oracle_py as op
            if op.cmult_regime(w, t, xser == "NEW") is None:
                continue
            ins, outs = [], []
            for _ in range(48):
                v = [R.operand(rng, w), R.operand(rng, w), R.operand(rng, t), R.operand(rng, t)]
                r = R.evaluate("int_cmult_dsp48", {"dtw": w, "twd": t, "xser": xser.lower()},
                               {"di_re": tw.vec(v[0], w), "di_im": tw.vec(v[1], w), "ww_re": tw.vec(v[2], t), "ww_im": tw.vec(v[3], t)})
                ins.append(v)
                outs.append([tw.signed(r["do_re"], w), tw.signed(r["do_im"], w)])
            out["cm_%d_%d_%s_in" % (w, t, xser)] = np.array(ins, dtype=np.int64)
            out["cm_%d_%d_%s_out" % (w, t, xser)] = np.array(outs, dtype=np.int64)
        for kind in ("dif", "dit"):
            for (dtw, tfw, scale, rnd, stage, odd) in FLY:
                g = {"stage": stage, "scale": scale, "dtw": dtw, "tfw": tfw, "rndmode": rnd, "xser": xser.lower()}
                wo = dtw - scale + 1
                ins, outs = [], []
                for _ in range(32):
                    v = [R.operand(rng, dtw) for _ in range(4)] + [R.operand(rng, tfw), R.operand(rng, tfw)]
                    i = {"ia_re": tw.vec(v[0], dtw), "ia_im": tw.vec(v[1], dtw), "ib_re": tw.vec(v[2], dtw), "ib_im": tw.vec(v[3], dtw),
                         "ww_re": tw.vec(v[4], tfw), "ww_im": tw.vec(v[5], tfw), "in_en": 1, "rst": 0}
                    if stage == 1:
                        i["dt_sw"] = odd
                    r = R.evaluate("int_%s2_fly" % kind, g, i)
                    ins.append(v)
                    outs.append([tw.signed(r[k], wo) for k in ("oa_re", "oa_im", "ob_re", "ob_im")])
                key = "fly_%s_w%d_t%d_s%d_r%d_st%d_o%d_%s" % (kind, dtw, tfw, scale, rnd, stage, odd, xser)
                out[key + "_in"] = np.array(ins, dtype=np.int64)
                out[key + "_out"] = np.array(outs, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "from_rtl_text.npz"), **out)
    print("wrote from_rtl_text.npz: %d arrays, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "from_rtl_text.npz"))))


if __name__ == "__main__":
    main()
