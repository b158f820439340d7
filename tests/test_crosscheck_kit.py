"""The external-pin kit (tools/vivado_crosscheck/): its committed expectations were produced by the GPU engine; here
(CPU) they are checked against the oracle, and compare.py is exercised on a passing and on a failing dump."""
import json
import os
import subprocess
import sys

import numpy as np

from intfftk_amd import textio
from oracle import oracle_c as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT = os.path.join(ROOT, "tools", "vivado_crosscheck")
EXP = os.path.join(KIT, "expected")


def _cases():
    return json.load(open(os.path.join(EXP, "manifest.json")))["cases"]


def test_kit_expectations_equal_the_oracle():
    cases = _cases()
    assert {(c["case"], c["mode"]) for c in cases} >= {("single_n7", "TRUNCATE"), ("single_n7", "ROUNDING"), ("single_n7", "UNSCALED"),
                                                       ("single_n12", "TRUNCATE"), ("pair_n7", "UNSCALED"),
                                                       ("single_n7_w24t24", "UNSCALED"), ("single_n7_w30t16", "TRUNCATE"),
                                                       ("single_n7_w30t16old", "TRUNCATE"), ("single_n7_w20t24", "ROUNDING"),
                                                       ("single_n7_w14t24", "UNSCALED")}
    regimes = set()
    for c in cases:  # the kit reaches every multiplier family that fits a VHDL integer
        for ii in range(c["nfft"]):
            w = c.get("data_width", 16) + ii * c["format"] + c["format"]
            regimes.add(C.cmult_regime(w, c.get("twdl_width", 16), c.get("xser", "NEW") == "NEW"))
    assert {"sngl", "dbl18", "sngl25", "dbl35", "trpl18", "trpl52"} <= regimes  # round 6: the hex testbenches carry the wide families too
    wide = [c for c in cases if c["out_bits"] > 32]
    assert len(wide) >= 6 and all(c.get("text") == "hex" and c["tb"] in ("tb_single_hex", "tb_pair_hex") for c in wide)
    assert {(c["data_width"], c["twdl_width"], c["xser"]) for c in wide} >= {(46, 16, "NEW"), (44, 16, "OLD"), (40, 24, "NEW"), (30, 24, "NEW"),
                                                                             (30, 16, "NEW"), (30, 16, "OLD"), (28, 24, "NEW")}
    # the dbl18 pair shares ONE stimulus, and the two series give different results on it (XSER-divergent vector)
    new, old = (next(c for c in cases if c["case"] == k) for k in ("hex_n7_w30t16", "hex_n7_w30t16old"))
    assert new["stimulus"] == old["stimulus"]
    assert not np.array_equal(textio.read_hex(os.path.join(EXP, new["expected"]), 37), textio.read_hex(os.path.join(EXP, old["expected"]), 37))
    for c in cases:
        n = 1 << c["nfft"]
        p = C.make_params(c["nfft"], c.get("data_width", 16), c.get("twdl_width", 16), c["format"], c["rndmode"], c.get("xser", "NEW") == "NEW")
        if c.get("text") == "hex":
            xt = textio.read_hex(os.path.join(EXP, c["stimulus"]), c["data_width"])
            gt = textio.read_hex(os.path.join(EXP, c["expected"]), c["out_bits"])
            if c["tb"] == "tb_single_hex":
                x, got, want = xt.reshape(-1, n, 2), gt.reshape(-1, n, 2), None
                want = C.execute(x, p, C.FWD)
            else:
                x, got = textio.table_to_double(xt, n), textio.table_to_double(gt, n)
                want = C.execute(x, p, C.PAIR)
        elif c["tb"] == "tb_single_dump":
            x = textio.read_di_single(os.path.join(EXP, c["stimulus"]), n)
            got = textio.read_di_single(os.path.join(EXP, c["expected"]), n)
            want = C.execute(x, p, C.FWD)
        else:
            x = textio.read_di_double(os.path.join(EXP, c["stimulus"]), n)
            a = np.loadtxt(os.path.join(EXP, c["expected"]), dtype=np.int64, ndmin=2)  # Q0_RE Q1_RE Q0_IM Q1_IM per beat
            got = np.stack([np.stack([a[:, 0], a[:, 2]], -1), np.stack([a[:, 1], a[:, 3]], -1)], 1).reshape(-1, n, 2)
            want = C.execute(x, p, C.PAIR)
        assert x.shape[0] == c["frames"]
        assert np.array_equal(got, want), (c["case"], c["mode"])


def test_compare_script_pass_and_fail(tmp_path):
    cmp_py = os.path.join(KIT, "compare.py")
    good = os.path.join(EXP, "single_n7_expected_ROUNDING.dat")
    r = subprocess.run([sys.executable, cmp_py, "single_n7", "ROUNDING", good], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("PASS")
    a = np.loadtxt(good, dtype=np.int64, ndmin=2)
    a[1000, 1] += 1
    bad = tmp_path / "bad.dat"
    np.savetxt(bad, a, fmt="%d")
    r = subprocess.run([sys.executable, cmp_py, "single_n7", "ROUNDING", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "line 1001" in r.stdout and "frame 7, position 104" in r.stdout
    # the pair: an RTL dump carries the reference's slice mix-up (Q0_IM = re of lane 0, Q1_RE = im of lane 1)
    e = np.loadtxt(os.path.join(EXP, "pair_n7_expected_UNSCALED.dat"), dtype=np.int64, ndmin=2)
    rtl = e.copy()
    rtl[:, 2], rtl[:, 1] = e[:, 0], e[:, 3]
    dump = tmp_path / "pair.dat"
    np.savetxt(dump, rtl, fmt="%d")
    assert subprocess.run([sys.executable, cmp_py, "pair_n7", "UNSCALED", str(dump)]).returncode == 0
    assert subprocess.run([sys.executable, cmp_py, "pair_n7", "UNSCALED", str(dump), "--no-reference-wiring"],
                          capture_output=True).returncode == 1
    # the strobe corner (tools/rtl_sim.py): a case that is PREDICTED TO DIFFER passes on a differing dump and fails on the engine's own
    exp = os.path.join(EXP, "hex_n7_w47t16_strobe_expected_TRUNCATE.hex")
    r = subprocess.run([sys.executable, cmp_py, "hex_n7_w47t16_strobe", "TRUNCATE", exp], capture_output=True, text=True)
    assert r.returncode == 1 and "prediction does not hold" in r.stdout
    lines = open(exp).read().splitlines()
    lines[1500], lines[1501] = lines[1501], lines[1500]   # inside a random frame
    (tmp_path / "strobe.hex").write_text("\n".join(lines) + "\n")
    r = subprocess.run([sys.executable, cmp_py, "hex_n7_w47t16_strobe", "TRUNCATE", str(tmp_path / "strobe.hex")], capture_output=True, text=True)
    assert r.returncode == 0 and "as predicted" in r.stdout
    assert subprocess.run([sys.executable, cmp_py, "hex_n7_w49t16_strobe", "TRUNCATE",
                           os.path.join(EXP, "hex_n7_w49t16_strobe_expected_TRUNCATE.hex")], capture_output=True).returncode == 0


def test_kit_lint_against_the_reference_entities(tmp_path):
    """No VHDL front end in the build image: what CAN be checked statically is -- every generic / port the kit testbenches
    associate exists with that spelling and width in the reference's entities, no conv_integer operand exceeds a VHDL integer,
    stimulus and expectations fit their widths (tools/vivado_crosscheck/lint_kit.py).  Needs the reference tree: build
    container only."""
    import importlib.util

    import pytest

    ref = os.environ.get("INTFFT_REFERENCE_DIR", "/root/reference")
    if not os.path.exists(os.path.join(ref, "src", "vhdl", "main", "int_fft_single_path.vhd")):
        pytest.skip("reference tree not present (GPU box)")
    spec = importlib.util.spec_from_file_location("lint_kit", os.path.join(KIT, "lint_kit.py"))
    lk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lk)
    errs, man = lk.lint(ref)
    assert not errs, errs
    assert len(man["cases"]) == 27
    assert {c["case"]: c["predicted"] for c in man["cases"] if "predicted" in c} == {"hex_n7_w47t16_strobe": "differs", "hex_n7_w49t16_strobe": "equal"}
    for c in man["cases"]:  # every vector dumped through conv_integer fits a VHDL integer; the wider ones go through the hex testbenches
        assert c["out_bits"] <= 32 or c.get("text") == "hex", c
    # the lint does catch what it is there for: a misspelt port, a wrong width, a too-wide conv_integer
    for tbfile in ("tb_single_dump.vhd", "tb_pair_dump.vhd", "tb_single_hex.vhd", "tb_pair_hex.vhd"):
        (tmp_path / tbfile).write_text(open(os.path.join(KIT, tbfile)).read())
    t = (tmp_path / "tb_single_dump.vhd").read_text()
    (tmp_path / "tb_single_dump.vhd").write_text(t.replace("DO_VL   => do_vl", "DO_VAL  => do_vl")
                                                  .replace("std_logic_vector(OW-1 downto 0);", "std_logic_vector(OW+20 downto 0);", 1))
    errs, _ = lk.lint(ref, kit=str(tmp_path))
    assert any("DO_VAL is not a port" in e for e in errs) and any("DO_VL of int_fft_single_path is left unassociated" in e for e in errs)
    assert any("DO_RE is" in e and "signal do_re is" in e for e in errs) and any("conv_integer(DO_RE)" in e for e in errs)
    # ... and the hex testbenches' own failure modes: an hread operand that is not a whole number of hex digits, a stimulus slice that
    # does not match the port, a sign extension to the wrong size
    t = (tmp_path / "tb_single_hex.vhd").read_text()
    (tmp_path / "tb_single_hex.vhd").write_text(t.replace("constant IH     : integer := 4*((DATA_WIDTH+3)/4);", "constant IH     : integer := DATA_WIDTH+1;")
                                                 .replace("vi := SXT(do_im, OH);", "vi := SXT(do_im, OW);"))
    t = (tmp_path / "tb_pair_hex.vhd").read_text()
    (tmp_path / "tb_pair_hex.vhd").write_text(t.replace("d1_im <= d(DATA_WIDTH-1 downto 0);", "d1_im <= d(DATA_WIDTH-2 downto 0);"))
    errs, _ = lk.lint(ref, kit=str(tmp_path))
    assert any("hread / hwrite operand A is 47 bits" in e for e in errs)               # hex_n7_w46t16: 46 + 1
    assert any("VI := SXT(DO_IM" in e for e in errs)
    assert any("D1_IM is 24 bits but takes a 23-bit slice of D" in e for e in errs)
