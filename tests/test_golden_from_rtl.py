"""tests/golden/from_rtl_text.npz -- vectors computed from the reference's own VHDL text (tests/golden/make_golden_from_rtl.py runs
tools/rtl_interp.py where the reference tree exists) -- against the C oracle, the Python twin (CPU) and the HIP path (-m gpu).

These are the only fixtures of the repository whose provenance is the reference's text itself rather than a reading of it: twiddle
tables of STAGE 2 .. 10 come from executing rom_twiddle_int's own function and process (no DSP48 involved); STAGE 11 .. 18, the complex
multiplier and the butterflies add the DSP48 slice model of oracle/dsp48_twin.py.  The file travels to the GPU box; the reference does not."""
import os
import re

import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_py as P

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "from_rtl_text.npz"))
TW_FULL = sorted(k for k in G.files if re.match(r"tw_s\d+_t\d+_(NEW|OLD)$", k))
TW_SAMP = sorted(k[:-4] for k in G.files if re.match(r"tw_s\d+_t\d+_(NEW|OLD)_idx$", k))
CM = sorted(k[:-3] for k in G.files if k.startswith("cm_") and k.endswith("_in"))
FLY = sorted(k[:-3] for k in G.files if k.startswith("fly_") and k.endswith("_in"))


def test_fixture_covers_what_it_says():
    assert len(TW_FULL) == 2 * 2 * 9 and len(TW_SAMP) == 2 * 2 * 8 and len(CM) >= 24 and len(FLY) == 2 * 2 * 12


@pytest.mark.parametrize("key", TW_FULL + TW_SAMP)
def test_oracle_twiddles_equal_the_text(key):
    s, t, ser = re.match(r"tw_s(\d+)_t(\d+)_(NEW|OLD)", key).groups()
    s, t, new = int(s), int(t), ser == "NEW"
    c = np.stack(C.twiddles(s, t, new), axis=-1) if isinstance(C.twiddles(s, t, new), tuple) else np.asarray(C.twiddles(s, t, new))
    p = np.array(P.twiddles(s, t, new), dtype=np.int64)
    if key in G.files:
        want = G[key].astype(np.int64)
        assert np.array_equal(p, want) and np.array_equal(np.asarray(c, dtype=np.int64).reshape(-1, 2), want)
    else:
        idx, want = G[key + "_idx"], G[key + "_val"].astype(np.int64)
        assert np.array_equal(p[idx], want) and np.array_equal(np.asarray(c, dtype=np.int64).reshape(-1, 2)[idx], want)


@pytest.mark.parametrize("key", CM)
def test_oracle_cmult_equals_the_text(key):
    w, t, ser = re.match(r"cm_(\d+)_(\d+)_(NEW|OLD)", key).groups()
    w, t, new = int(w), int(t), ser == "NEW"
    for v, o in zip(G[key + "_in"], G[key + "_out"]):
        v = [int(x) for x in v]
        assert P.cmult(*v, w, t, new) == (int(o[0]), int(o[1]))
        if w <= 62:
            assert tuple(C.cmult(*v, w, t, new)) == (int(o[0]), int(o[1]))


@pytest.mark.parametrize("key", FLY)
def test_oracle_butterflies_equal_the_text(key):
    kind, dtw, tfw, scale, rnd, stage, odd, ser = re.match(r"fly_(dif|dit)_w(\d+)_t(\d+)_s(\d)_r(\d)_st(\d+)_o(\d)_(NEW|OLD)", key).groups()
    dtw, tfw, scale, rnd, stage, odd = (int(x) for x in (dtw, tfw, scale, rnd, stage, odd))
    f = P.dif_fly if kind == "dif" else P.dit_fly
    wo = dtw - scale + 1
    for v, o in zip(G[key + "_in"], G[key + "_out"]):
        v = [int(x) for x in v]
        r = f((v[0], v[1]), (v[2], v[3]), (v[4], v[5]), stage, dtw, tfw, scale, rnd, odd, ser == "NEW")
        assert tuple(P.sgn(x, wo) for x in (r[0][0], r[0][1], r[1][0], r[1][1])) == tuple(int(x) for x in o)


@pytest.mark.gpu
def test_hip_twiddles_equal_the_text():
    """intfft_twiddles of real plans (the tables the kernels multiply by, generated on the device by k_twiddle_stage)"""
    from intfftk_amd import IntFFTCore

    for t in (16, 24):
        for ser in ("NEW", "OLD"):
            core = IntFFTCore(19, 16, t, 0, 0, ser, "FWD")
            for s in range(2, 19):
                got = core.twiddles(s).astype(np.int64).reshape(-1, 2)
                key = "tw_s%d_t%d_%s" % (s, t, ser)
                if key in G.files:
                    assert np.array_equal(got, G[key].astype(np.int64)), key
                else:
                    assert np.array_equal(got[G[key + "_idx"]], G[key + "_val"].astype(np.int64)), key
            core.close()
