"""The generator of tests/test_gpu_fuzz.py is a pure function of its seed and reaches every axis (CPU: needs the oracle's validator only)."""
import pytest

from tests.test_gpu_fuzz import CHUNKS, PER_CHUNK, configurations


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_fuzz_generator_is_deterministic_and_covers_the_axes(chunk):
    a, b = configurations(0x5EED0500 + chunk, PER_CHUNK), configurations(0x5EED0500 + chunk, PER_CHUNK)
    assert a == b and len(a) == PER_CHUNK
    assert {c["d"] for c in a} == {"FWD", "INV", "PAIR"} and {c["fmt"] for c in a} == {0, 1} and any(c["rnd"] for c in a)
    assert any(c["in_o"] != "NATURAL" or c["out_o"] != "NATURAL" for c in a) and any(c["log2n"] >= 13 for c in a)


def test_fuzz_set_as_a_whole():
    allc = [c for k in range(CHUNKS) for c in configurations(0x5EED0500 + k, PER_CHUNK)]
    assert {c["log2n"] for c in allc} >= set(range(3, 21)) - {15, 17, 18, 19}  # every short length; the long ones are drawn rarely
    assert any(c["l1"] for c in allc) and any(c["dw"] > 32 for c in allc) and any(not c["new"] for c in allc)
    assert {c["in_o"] for c in allc} == {"NATURAL", "BITREV", "HALVES", "BITREV_LANES"}
    assert {c["use_fly"] for c in allc} == {0, 1}
