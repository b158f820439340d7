"""The reference's whole core, clocked from its own text (tools/rtl_sim.py), against the oracle -- this container only.

tools/rtl_interp.py pins the arithmetic entities one by one; this goes end to end: int_fftNk / int_ifftNk (int_fftNk.vhd / int_ifftNk.vhd)
are elaborated down to the DSP48 primitives with the pipeline registers their generic maps ask for, fed frames beat by beat, and the
DO_VAL-qualified output beats must be the oracle's frames.  What that adds to the per-entity checks: the latencies (aligners, valid
strobes, twiddle counters, delay lines against the butterflies' pipeline depth) line up in the text exactly where the oracle assumes they
do.  Small frames only (a simulated clock costs milliseconds); tools/rtl_sim.py --sweep is the longer list (profiles/r06_rtl_sim.txt).
Skipped where /root/reference is absent (the GPU box)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
import rtl_sim as S  # noqa: E402

pytestmark = pytest.mark.skipif(not S.available(), reason="the reference's text is not on this machine")


@pytest.mark.parametrize("direction", ["FWD", "INV"])
@pytest.mark.parametrize("mode", [(0, 0), (0, 1), (1, 0)], ids=["truncate", "round", "unscaled"])
def test_eight_point_core_every_mode(direction, mode):
    fmt, rnd = mode
    ok, got, want = S.compare(direction, 3, 16, 16, fmt, rnd, "NEW", "cont", count=2)
    assert ok and got == want == 8, (got, want)


def test_old_series_and_wrapped_delay_lines():
    # RAMB_TYPE = "WRAP": the lines move only while beats come in, the tail of the last frame stays inside -- the prefix must be right
    ok, got, want = S.compare("FWD", 3, 16, 16, 0, 1, "OLD", "wrap")
    assert ok and want - 4 <= got <= want, (got, want)
    ok, got, want = S.compare("INV", 3, 16, 16, 1, 0, "OLD", "wrap")
    assert ok and want - 4 <= got <= want, (got, want)


def test_sixteen_points_crosses_a_delay_line_of_two_beats():
    ok, got, want = S.compare("INV", 4, 16, 16, 0, 1, "NEW", "cont", count=2)
    assert ok and got == want == 16, (got, want)


def test_idle_clocks_between_frames_and_the_bypass_mux():
    assert S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", gap=5, count=2)[0]
    # USE_FLY = 0 (int_fftNk.vhd:260-277): the oracle's reading -- the butterflies drop out, the commutation stays
    assert S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", use_fly=0, count=2)[0]
    assert S.compare("INV", 3, 16, 16, 1, 0, "NEW", "cont", use_fly=0, count=2)[0]


@pytest.mark.parametrize("cfg", [("FWD", 24, 24, 1, 0, "NEW"), ("INV", 52, 16, 1, 0, "OLD")], ids=lambda c: "%s_w%d_t%d" % c[:3])
def test_wide_regimes_through_the_whole_core(cfg):
    # dbl35 / trpl18 multipliers and the two-slice adder (DSPW >= 48) inside the pipeline, latencies included
    d, dw, tw_, fmt, rnd, xser = cfg
    assert S.compare(d, 3, dw, tw_, fmt, rnd, xser, "cont", count=1)[0]


def test_the_simulation_notices_a_wrong_oracle(monkeypatch):
    from oracle import oracle_py as op
    real = op.fft_dif

    def off_by_one(fr, *a):
        v = list(real(fr, *a))
        v[5] = (v[5][0] + 1, v[5][1])
        return v
    monkeypatch.setattr(op, "fft_dif", off_by_one)
    assert not S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", count=2)[0]


def test_the_strobe_corner_of_the_reference():
    """A defect of the reference this simulation found (profiles/HISTORY.md): the butterflies take their valid-strobe
    delay from addsub_delay(DTW + SCALE + RNDMODE) but build the adder for DSPW = DTW - 1 (scaled truncate) / DTW (scaled round, unscaled)
    bits; at scaled DTW = 46 (round), 47, 48 (truncate) the two fall on different sides of the 48-bit slice boundary and the strobe leaves
    one clock off.  The text's frames are wrong there; one bit to either side they are the oracle's.  Outside the documented DATA_WIDTH range; the engine follows the arithmetic."""
    assert not S.compare("FWD", 3, 47, 16, 0, 0, "NEW", "cont", count=1)[0]
    assert S.compare("FWD", 3, 49, 16, 0, 0, "NEW", "cont", count=1)[0]    # (unscaled 47 / 48, where argument and DSPW are both DTW: the sweep)


def test_the_wrappers_with_their_buffers():
    """src/vhdl/main: int_fft_single_path (natural order in, natural order out through inbuf_half_path / outbuf_half_path /
    int_bitrev_order) and int_fft_ifft_pair (iobuf_flow_int2 around FFT -> IFFT; lanes = x[2i], x[2i + 1]): what the C-ABI calls NATURAL."""
    ok, whole, count = S.compare_wrapper("single", 3, 16, 16, 0, 1, "OLD", count=3, flush=False)
    assert ok and whole == count - 1, (whole, count)   # the bit-reverse buffer keeps the last frame until another one comes
    ok, whole, count = S.compare_wrapper("single", 5, 16, 16, 0, 1, "OLD", count=3)
    assert ok and whole == count, (whole, count)       # ... an all-zero frame behind the data pushes it out
    ok, whole, count = S.compare_wrapper("pair", 3, 16, 16, 0, 0, "NEW", count=3)
    assert ok and whole == count, (whole, count)


def test_the_simulation_follows_the_timing_of_the_text(monkeypatch):
    """Two edits of the TEXT that change no arithmetic, only WHEN things happen: the frames must go wrong."""
    import rtl_interp as R
    real = R._load

    def edited(old, new):
        def load(entity):
            return real(entity).replace(old, new)
        return load

    assert S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", count=2)[0]
    # 1. the butterfly's valid strobe one clock late (the mechanism of the reference's own defect at 47 bits, seeded at 16)
    assert "addsub_delay(dtw+scale+rndmode)" in real("int_dif2_fly")
    monkeypatch.setattr(R, "_load", edited("addsub_delay(dtw+scale+rndmode)", "addsub_delay(dtw+scale+rndmode+40)"))
    R.forget()
    assert not S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", count=2)[0]
    # 2. the twiddle enable through one more register: the twiddles arrive a beat after their data
    assert "tw_en <= bf_en;" in real("int_align_fft")
    monkeypatch.setattr(R, "_load", edited("tw_en <= bf_en;", "tw_en <= bf_en when rising_edge(clk);"))
    R.forget()
    assert not S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", count=2)[0]
    monkeypatch.setattr(R, "_load", real)
    R.forget()
    assert S.compare("FWD", 3, 16, 16, 0, 0, "NEW", "cont", count=2)[0]


def test_longer_frames_and_the_wrap_pair():
    assert S.compare("FWD", 6, 16, 16, 0, 1, "NEW", "cont", count=2)[0]
    assert S.compare("INV", 7, 16, 16, 0, 0, "OLD", "wrap", count=3)[0]
    # int_fft_ifft_pair as the reference's fft_double_test.vhd drives it: RAMB_TYPE = "WRAP" (iobuf_wrap_int2, ramb_tdp_rw), the enable of
    # every beat followed by an idle clock, 32 idle clocks between frames; four all-zero frames push the data frames out
    import random
    frames = S._frames(random.Random(9), 4, 16, 3)
    beats, _ = S.run_pair(4, 16, 16, 0, 0, "NEW", frames, "wrap", gap=32, flush=4, toggle=True)
    got = [s for b in beats for s in b]
    want = S.expected_natural("PAIR", 4, 16, 16, 0, 0, "NEW", frames)
    assert len(got) >= len(want) and got[:len(want)] == want


def test_the_cross_check_kit_on_this_simulation(capsys):
    """One case of tools/vivado_crosscheck through the text with the testbench's protocol, judged by the kit's own compare.py
    (all of them: python tools/rtl_sim.py --kit, profiles/r06_rtl_sim_kit.txt)."""
    assert S.run_kit({"single_n7_w14t16"}) == 0
    assert "PASS  single_n7_w14t16 ROUNDING" in capsys.readouterr().out


def test_random_generics_through_the_whole_core():
    """A few draws of tools/rtl_sim.py --fuzz (profiles/r06_rtl_sim_fuzz.txt holds 500): random NFFT / widths / mode / series / direction /
    RAMB_TYPE, the text clocked against the oracle; what the oracle's validator accepts must elaborate."""
    assert S.fuzz(8, 12345) == 0
