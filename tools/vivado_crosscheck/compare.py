#!/usr/bin/env python3
"""PASS / FAIL of an RTL simulation against the GPU engine's expected dumps (pure numpy: runs anywhere).

    python compare.py <case> <mode> <rtl_dump.dat> [--expected-dir expected] [--no-reference-wiring]

<case> / <mode> name an entry of expected/manifest.json (e.g. single_n7 TRUNCATE).  For the pair the reference wires
Q0_IM / Q1_RE to the wrong slices (int_fft_ifft_pair.vhd:332-335: Q0_IM carries the REAL part of lane 0, Q1_RE the
IMAGINARY part of lane 1), so by default the expected beats are re-wired the same way before comparing; pass
--no-reference-wiring for a dump of a corrected RTL.  A dump may hold more frames than the expectation (the RTL keeps
emitting after the last frame in some modes): only whole leading frames are compared, and a shorter dump FAILS.
A case whose manifest entry says "predicted": "differs" (the strobe corner of the reference found by tools/rtl_sim.py) PASSES when the
RTL's dump is NOT the engine's.  Exit status 0 = PASS."""
import argparse
import json
import os
import sys

import numpy as np


def read_hex(path):
    """[lines, columns] int64 of a hex dump: every word is two's complement of its own digit count (the testbenches sign-extend to the
    digit boundary), so leading-digit conventions of the simulator's hwrite do not matter as long as the sign is carried"""
    rows = []
    for line in open(path):
        f = line.split()
        if f:
            vals = []
            for w in f:
                v, bits = int(w, 16), 4 * len(w)
                vals.append(v - (1 << bits) if v >> (bits - 1) else v)
            rows.append(vals)
    return np.array(rows, dtype=np.int64).reshape(len(rows), -1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("mode")
    ap.add_argument("dump")
    ap.add_argument("--expected-dir", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "expected"))
    ap.add_argument("--no-reference-wiring", action="store_true")
    a = ap.parse_args()
    man = json.load(open(os.path.join(a.expected_dir, "manifest.json")))
    ent = [c for c in man["cases"] if c["case"] == a.case and c["mode"] == a.mode]
    if not ent:
        sys.exit("no such case/mode in the manifest: %s %s" % (a.case, a.mode))
    ent = ent[0]
    pair = ent["tb"] in ("tb_pair_dump", "tb_pair_hex")
    if ent.get("text") == "hex":  # full-width two's-complement hex words (tb_single_hex / tb_pair_hex)
        want, got = read_hex(os.path.join(a.expected_dir, ent["expected"])), read_hex(a.dump)
    else:
        want = np.loadtxt(os.path.join(a.expected_dir, ent["expected"]), dtype=np.int64, ndmin=2)
        got = np.loadtxt(a.dump, dtype=np.int64, ndmin=2)
    if pair and not a.no_reference_wiring:
        want = want.copy()
        want[:, 2] = want[:, 0]  # Q0_IM <- re of lane 0
        want[:, 1] = want[:, 3]  # Q1_RE <- im of lane 1   (columns: Q0_RE Q1_RE Q0_IM Q1_IM)
    if got.shape[1] != want.shape[1]:
        sys.exit("FAIL: %d columns in the dump, %d expected" % (got.shape[1], want.shape[1]))
    if got.shape[0] < want.shape[0]:
        sys.exit("FAIL: the dump holds %d lines, %d expected (simulation stopped early?)" % (got.shape[0], want.shape[0]))
    got = got[: want.shape[0]]
    bad = np.argwhere(got != want)
    per_frame = (1 << ent["nfft"]) // (2 if pair else 1)
    if ent.get("predicted") == "differs":
        # the simulation of the reference's own text (tools/rtl_sim.py) says this RTL mis-times its valid strobe here: a difference CONFIRMS it
        if len(bad):
            print("PASS  %s %s: the RTL differs from the arithmetic in %d of %d values, as predicted (%s)" % (a.case, a.mode, len(bad), want.size, ent.get("note", "")))
            return 0
        print("FAIL  %s %s: the RTL is bit-exact here, but a difference was predicted (%s): the prediction does not hold" % (a.case, a.mode, ent.get("note", "")))
        return 1
    if len(bad) == 0:
        print("PASS  %s %s: %d lines (%d frames of 2^%d points) bit-exact" % (a.case, a.mode, want.shape[0], want.shape[0] // per_frame, ent["nfft"]))
        return 0
    r, c = bad[0]
    print("FAIL  %s %s: %d of %d values differ; first at line %d (frame %d, position %d) column %d: RTL %d, engine %d; max |diff| %d"
          % (a.case, a.mode, len(bad), want.size, r + 1, r // per_frame, r % per_frame, c, got[r, c], want[r, c], np.abs(got - want).max()))
    return 1


if __name__ == "__main__":
    sys.exit(main())
