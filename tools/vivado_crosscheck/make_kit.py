#!/usr/bin/env python3
"""Generates the stimulus files and the engine's expected dumps of the external-pin kit (needs a HIP device).

    python tools/vivado_crosscheck/make_kit.py [--out tools/vivado_crosscheck/expected]

Cases (each: one stimulus file + one expected dump per mode):
  single_n7     NFFT = 7  (the testbenches' own length, fft_signle_test.vhd:93): 8 edge frames + 24 random frames
  single_n12    NFFT = 12 (first length whose STAGE 11 twiddles come from row_twiddle_tay): impulse at n = 1 (reads the
                twiddle tables out), full-scale random, chirp
  pair_n7       NFFT = 7 int_fft_ifft_pair: di_double.dat beats, full-width expected output for FORMAT 0 and 1
  single_n7_w24t24 / _w30t16 / _w30t16old / _w20t24 / _w14t24   NFFT = 7 at other widths, so that the other multiplier regimes meet the RTL too:
                24 x 24 unscaled (sngl25 -> dbl35, 31-bit results), 30 x 16 scaled (dbl18; XSER NEW and OLD differ there),
                20 x 24 scaled (dbl35), 14 x 24 unscaled (sngl25 -> dbl35).  Every value still fits a VHDL integer, which is what the testbench reads and writes.
  hex_*         round 6: the same two testbench shapes with HEXADECIMAL full-width text I/O (tb_single_hex.vhd / tb_pair_hex.vhd,
                ieee.std_logic_textio hread / hwrite), for what a VHDL integer cannot carry: trpl18 in both series (46 x 16 NEW, 44 x 16
                OLD: int_cmult_trpl18_dsp48.vhd:151-162), trpl52 (40 x 24 unscaled, 38 x 24 rounding: int_cmult_trpl52_dsp48.vhd:166-170),
                dbl35 at 30 .. 36 bits (30 x 24 unscaled, 35 x 24 truncate: int_cmult_dbl35_dsp48.vhd:163-168), the XSER-divergent dbl18
                pair 30 x 16 unscaled NEW / OLD (37-bit results), NFFT = 12 at 28 x 24 unscaled (the dbl35 -> trpl52 walk of BASELINE
                config 3 + the Taylor twiddles of STAGE 11, 40-bit results), and a 24 x 16 unscaled pair (38-bit results)
  hex_n7_w47t16_strobe / hex_n7_w49t16_strobe   a falsifiable prediction of tools/rtl_sim.py (the reference's text clocked cycle by cycle): at
                scaled DATA_WIDTH 47 the RTL's own valid strobe is one clock off and its frames DIFFER from the arithmetic; at 49 they agree
Modes: TRUNCATE (FORMAT 0, RNDMODE 0), ROUNDING (0, 1), UNSCALED (1, 0)  -- fft_signle_test.vhd:80-112.
Everything is produced by the GPU engine through the C-ABI (intfftk_amd); the oracle is not involved.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
MODES = {"TRUNCATE": (0, 0), "ROUNDING": (0, 1), "UNSCALED": (1, 0)}


def frames_for(nfft, which):
    from tests.helpers import chirp_frame, edge_frames, uniform_frames

    n = 1 << nfft
    if which == "full":
        return np.concatenate([edge_frames(n, 16), uniform_frames(24, n, 16, 0x1F7 + nfft)])
    imp = np.zeros((1, n, 2), dtype=np.int64)
    imp[0, 1, 0] = 1 << 14
    return np.concatenate([imp, uniform_frames(1, n, 16, 0xBEEF), chirp_frame(n)[None] * 64])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tools", "vivado_crosscheck", "expected"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    import torch

    from intfftk_amd import int_fft_ifft_pair, int_fft_single_path, textio

    manifest = {"generated_by": "intfftk_amd GPU engine (libintfft.so)", "cases": []}
    for name, nfft, which in [("single_n7", 7, "full"), ("single_n12", 12, "taylor")]:
        x = frames_for(nfft, which)
        textio.write_di_single(os.path.join(a.out, "%s_di_single.dat" % name), x)
        for mode, (fmt, rnd) in MODES.items():
            core = int_fft_single_path(nfft, 16, 16, fmt, rnd, "NEW")
            y = core(torch.from_numpy(x.astype(np.int16)).cuda()).cpu().numpy()
            assert core.out_bits <= 32, "the testbench dumps through conv_integer: a VHDL integer holds 32 bits"
            textio.write_di_single(os.path.join(a.out, "%s_expected_%s.dat" % (name, mode)), y)
            manifest["cases"].append({"case": name, "tb": "tb_single_dump", "nfft": nfft, "mode": mode, "format": fmt, "rndmode": rnd,
                                      "frames": int(x.shape[0]), "stimulus": "%s_di_single.dat" % name,
                                      "expected": "%s_expected_%s.dat" % (name, mode), "out_bits": core.out_bits})
            core.close()
    for name, dw, tw, mode, xser in [("single_n7_w24t24", 24, 24, "UNSCALED", "NEW"), ("single_n7_w30t16", 30, 16, "TRUNCATE", "NEW"),
                                     ("single_n7_w30t16old", 30, 16, "TRUNCATE", "OLD"), ("single_n7_w20t24", 20, 24, "ROUNDING", "NEW"),
                                     ("single_n7_w14t24", 14, 24, "UNSCALED", "NEW"),
                                     # narrow data on the packed int16 kernels (12- / 14-bit converters), both scaled modes
                                     ("single_n7_w12t16", 12, 16, "TRUNCATE", "NEW"), ("single_n7_w14t16", 14, 16, "ROUNDING", "NEW")]:
        from tests.helpers import edge_frames, uniform_frames

        fmt, rnd = MODES[mode]
        x = np.concatenate([edge_frames(128, dw), uniform_frames(8, 128, dw, 0x5EED + dw)])
        textio.write_di_single(os.path.join(a.out, "%s_di_single.dat" % name), x)
        core = int_fft_single_path(7, dw, tw, fmt, rnd, xser)
        dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
        y = core(torch.from_numpy(x.astype(dt)).cuda()).cpu().numpy()
        assert core.out_bits <= 32
        textio.write_di_single(os.path.join(a.out, "%s_expected_%s.dat" % (name, mode)), y)
        manifest["cases"].append({"case": name, "tb": "tb_single_dump", "nfft": 7, "mode": mode, "format": fmt, "rndmode": rnd,
                                  "data_width": dw, "twdl_width": tw, "xser": xser, "frames": int(x.shape[0]),
                                  "stimulus": "%s_di_single.dat" % name, "expected": "%s_expected_%s.dat" % (name, mode),
                                  "out_bits": core.out_bits})
        core.close()
    x = frames_for(7, "full")
    textio.write_di_double(os.path.join(a.out, "pair_n7_di_double.dat"), x)
    for mode, (fmt, rnd) in [("TRUNCATE", (0, 0)), ("UNSCALED", (1, 0))]:
        core = int_fft_ifft_pair(7, 16, 16, fmt, rnd, "NEW")
        y = core(torch.from_numpy(x.astype(np.int16)).cuda()).cpu().numpy().astype(np.int64)
        assert core.out_bits <= 32, "the testbench dumps through conv_integer: a VHDL integer holds 32 bits"
        beats = y.reshape(-1, 2, 2)  # [beat, lane, (re, im)]: Q0_RE Q1_RE Q0_IM Q1_IM as the ports SHOULD carry them
        np.savetxt(os.path.join(a.out, "pair_n7_expected_%s.dat" % mode),
                   np.stack([beats[:, 0, 0], beats[:, 1, 0], beats[:, 0, 1], beats[:, 1, 1]], axis=-1), fmt="%d")
        manifest["cases"].append({"case": "pair_n7", "tb": "tb_pair_dump", "nfft": 7, "mode": mode, "format": fmt, "rndmode": rnd,
                                  "frames": int(x.shape[0]), "stimulus": "pair_n7_di_double.dat",
                                  "expected": "pair_n7_expected_%s.dat" % mode, "out_bits": core.out_bits})
        core.close()
    # ---- full-width cases (hex text, tb_single_hex / tb_pair_hex) ------------------------------------------------------------
    from tests.helpers import chirp_frame, edge_frames, uniform_frames

    wide_single = [("hex_n7_w46t16", 7, 46, 16, "UNSCALED", "NEW"), ("hex_n7_w44t16old", 7, 44, 16, "UNSCALED", "OLD"),
                   ("hex_n7_w40t24", 7, 40, 24, "UNSCALED", "NEW"), ("hex_n7_w38t24", 7, 38, 24, "ROUNDING", "NEW"),
                   ("hex_n7_w30t24", 7, 30, 24, "UNSCALED", "NEW"), ("hex_n7_w35t24", 7, 35, 24, "TRUNCATE", "NEW"),
                   ("hex_n7_w30t16", 7, 30, 16, "UNSCALED", "NEW"), ("hex_n7_w30t16old", 7, 30, 16, "UNSCALED", "OLD"),
                   ("hex_n12_w28t24", 12, 28, 24, "UNSCALED", "NEW")]
    # A prediction of tools/rtl_sim.py (the reference's text, clocked): at scaled DTW 47 the butterflies' valid strobe is one clock off
    # (ADD_DELAY = addsub_delay(DTW+SCALE+RNDMODE)+RNDMODE against an adder of DTW-1 bits, int_dif2_fly.vhd), so the RTL's frames must DIFFER
    # from the arithmetic the engine computes; two bits up (49) the two agree again and the RTL must be bit-exact.  compare.py knows.
    predicted = {"hex_n7_w47t16_strobe": "differs", "hex_n7_w49t16_strobe": "equal"}
    wide_single += [("hex_n7_w47t16_strobe", 7, 47, 16, "TRUNCATE", "NEW"), ("hex_n7_w49t16_strobe", 7, 49, 16, "TRUNCATE", "NEW")]
    for name, nfft, dw, tw, mode, xser in wide_single:
        fmt, rnd = MODES[mode]
        n = 1 << nfft
        if nfft == 7:
            x = np.concatenate([edge_frames(n, dw), uniform_frames(8, n, dw, 0x5EED + dw)])
        else:  # impulse at n = 1 (reads the twiddle tables out), full-scale random, the chirp scaled up
            imp = np.zeros((1, n, 2), dtype=np.int64)
            imp[0, 1, 0] = 1 << (dw - 2)
            x = np.concatenate([imp, uniform_frames(1, n, dw, 0xBEEF), chirp_frame(n)[None] * (1 << (dw - 10))])
        stim = "%s_di_single.hex" % name.replace("hex_n7_w30t16old", "hex_n7_w30t16")  # NEW / OLD of the dbl18 pair share one stimulus
        textio.write_hex(os.path.join(a.out, stim), textio.single_to_table(x), dw)
        core = int_fft_single_path(nfft, dw, tw, fmt, rnd, xser)
        dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
        y = core(torch.from_numpy(x.astype(dt)).cuda()).cpu().numpy()
        assert core.out_container <= 8, "the hex dumps hold up to 64 bits"
        exp = "%s_expected_%s.hex" % (name, mode)
        textio.write_hex(os.path.join(a.out, exp), textio.single_to_table(y), core.out_bits)
        manifest["cases"].append({"case": name, "tb": "tb_single_hex", "text": "hex", "nfft": nfft, "mode": mode, "format": fmt, "rndmode": rnd,
                                  "data_width": dw, "twdl_width": tw, "xser": xser, "frames": int(x.shape[0]), "stimulus": stim,
                                  "expected": exp, "out_bits": core.out_bits, "kernel": core.info["kernel_name"]})
        if name in predicted:
            manifest["cases"][-1]["predicted"] = predicted[name]
            manifest["cases"][-1]["note"] = ("tools/rtl_sim.py: valid strobe of the butterflies one clock off at scaled DTW 46 (round), 47, 48 "
                                             "(truncate) -- the RTL is expected to %s here" % ("DIFFER" if predicted[name] == "differs" else "agree"))
        core.close()
    for name, nfft, dw, tw, mode, xser in [("hex_pair_n7_w24t16", 7, 24, 16, "UNSCALED", "NEW")]:
        fmt, rnd = MODES[mode]
        n = 1 << nfft
        x = np.concatenate([edge_frames(n, dw), uniform_frames(8, n, dw, 0xA11 + dw)])
        stim = "%s_di_double.hex" % name
        textio.write_hex(os.path.join(a.out, stim), textio.double_to_table(x), dw)
        core = int_fft_ifft_pair(nfft, dw, tw, fmt, rnd, xser)
        dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
        y = core(torch.from_numpy(x.astype(dt)).cuda()).cpu().numpy().astype(np.int64)
        exp = "%s_expected_%s.hex" % (name, mode)
        textio.write_hex(os.path.join(a.out, exp), textio.double_to_table(y), core.out_bits)
        manifest["cases"].append({"case": name, "tb": "tb_pair_hex", "text": "hex", "nfft": nfft, "mode": mode, "format": fmt, "rndmode": rnd,
                                  "data_width": dw, "twdl_width": tw, "xser": xser, "frames": int(x.shape[0]), "stimulus": stim,
                                  "expected": exp, "out_bits": core.out_bits, "kernel": core.info["kernel_name"]})
        core.close()
    with open(os.path.join(a.out, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print("wrote %d cases to %s" % (len(manifest["cases"]), a.out))


if __name__ == "__main__":
    main()
