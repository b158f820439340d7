"""Committed fixtures (tests/golden/, made by make_golden.py from the independent Python twin, plus the
survey-derived KATs) against the C oracle on CPU and against the HIP path (-m gpu)."""
import json
import os
import re
import zlib

import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_py as P

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = np.load(os.path.join(HERE, "frames.npz"))
TWID = np.load(os.path.join(HERE, "twiddles.npz"))
KATS = json.load(open(os.path.join(HERE, "kats.json")))
MODES = {"TRUNCATE": (0, 0), "ROUNDING": (0, 1), "UNSCALED": (1, 0)}
DIRS = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
ORD = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
CASES = sorted(k[:-3] for k in FRAMES.files if k.endswith("_in"))


def parse(key):
    m = re.match(r"n(\d+)_(\w+?)_(FWD|INV|PAIR)_w(\d+)_t(\d+)_(NEW|OLD)$", key)
    n, mode, d, w, t, ser = m.groups()
    fmt, rnd = MODES[mode]
    return int(n).bit_length() - 1, int(w), int(t), fmt, rnd, ser == "NEW", d


def test_fixture_inventory():
    assert len(CASES) >= 40
    assert {parse(k)[0] for k in CASES} >= {3, 4, 7, 10}


@pytest.mark.parametrize("key", CASES)
def test_c_oracle_matches_golden_frames(key):
    log2n, dw, tw, fmt, rnd, new, d = parse(key)
    p = C.make_params(log2n, dw, tw, fmt, rnd, new)
    for form in (0, 1):
        got = C.execute(FRAMES[key + "_in"], p, DIRS[d], form=form)
        assert np.array_equal(got, FRAMES[key + "_out"]), (key, form)


def _twiddle_keys():
    return sorted({re.match(r"(s\d+_t\d+_(?:NEW|OLD))", k).group(1) for k in TWID.files})


@pytest.mark.parametrize("key", _twiddle_keys())
def test_c_oracle_matches_golden_twiddles(key):
    s, t, ser = re.match(r"s(\d+)_t(\d+)_(NEW|OLD)", key).groups()
    re_, im_ = C.twiddles(int(s), int(t), ser == "NEW")
    got = np.stack([re_, im_], axis=-1).astype(np.int32)
    if key in TWID.files:
        assert np.array_equal(got, TWID[key])
    else:
        assert np.array_equal(got[TWID[key + "_idx"]], TWID[key + "_val"])
        assert zlib.crc32(got.tobytes()) == int(TWID[key + "_crc"][0])


def test_survey_kats_twiddles_and_cmult():
    for k in KATS["twiddles"]:
        re_, im_ = C.twiddles(k["stage"], k["t"], k["new"])
        got = [[int(a), int(b)] for a, b in zip(re_[:len(k["first"])], im_[:len(k["first"])])]
        assert got == k["first"]
        assert [list(v) for v in P.twiddles(k["stage"], k["t"], k["new"])[:len(k["first"])]] == k["first"]
    for dre, dim, wr, wi, w, t, new, regime, ore, oim in KATS["cmult"]:
        assert C.cmult_regime(w, t, new) == regime == P.cmult_regime(w, t, new)
        assert C.cmult(dre, dim, wr, wi, w, t, new) == (ore, oim) == P.cmult(dre, dim, wr, wi, w, t, new)


@pytest.mark.parametrize("i", range(len(KATS["frames"])))
def test_survey_kats_frames(i):
    k = KATS["frames"][i]
    p = C.make_params(k["log2n"], k["dw"], k["tw"], k["fmt"], k["rnd"], True)
    x = np.array(k["in"], dtype=np.int64)[None]
    got = C.execute(x, p, DIRS[k["dir"]], ORD[k.get("in_order", "NATURAL")])[0]
    assert got.tolist() == k["out"]


# ---- the same fixtures through the HIP path -----------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("key", CASES)
def test_hip_matches_golden_frames(key):
    import torch

    from intfftk_amd import IntFFTCore

    log2n, dw, tw, fmt, rnd, new, d = parse(key)
    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", d)
    dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
    y = core(torch.from_numpy(FRAMES[key + "_in"].astype(dt)).cuda()).cpu().numpy().astype(np.int64)
    assert np.array_equal(y, FRAMES[key + "_out"]), key
    core.close()


@pytest.mark.gpu
def test_hip_matches_golden_twiddles_and_kats():
    from intfftk_amd import IntFFTCore

    for t in (16, 24):
        for ser in ("NEW", "OLD"):
            core = IntFFTCore(20, 16, t, 0, 0, ser, "FWD")
            for s in range(20):
                key = "s%d_t%d_%s" % (s, t, ser)
                got = core.twiddles(s)
                if key in TWID.files:
                    assert np.array_equal(got, TWID[key]), key
                else:
                    assert np.array_equal(got[TWID[key + "_idx"]], TWID[key + "_val"]), key
                    assert zlib.crc32(got.tobytes()) == int(TWID[key + "_crc"][0]), key
            core.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(KATS["frames"])))
def test_hip_survey_kats_frames(i):
    import torch

    from intfftk_amd import IntFFTCore

    k = KATS["frames"][i]
    core = IntFFTCore(k["log2n"], k["dw"], k["tw"], k["fmt"], k["rnd"], "NEW", k["dir"], k.get("in_order", "NATURAL"))
    dt = {2: np.int16, 4: np.int32, 8: np.int64}[core.in_container]
    y = core(torch.from_numpy(np.array(k["in"], dtype=dt)[None]).cuda()).cpu().numpy()[0]
    assert y.tolist() == k["out"]
    core.close()
