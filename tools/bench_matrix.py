"""Coverage matrix: which kernel serves which (N, mode, direction) and how fast (one MI355X, data resident).
Prints a markdown table; every row is also checked against the oracle on a frame prefix (bench_configs.run)."""
import json
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tools import bench_configs as B  # noqa: E402

ROWS = []
for L in (3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 20):
    ROWS.append(("%d:16:16:0" % L, "16-bit scaled-trunc FWD"))
for L in (7, 10, 12):
    ROWS.append(("%d:16:16:0:1" % L, "16-bit scaled-round FWD"))
for L in (10, 12):
    ROWS.append(("%d:16:16:0:1:INV" % L, "16-bit scaled-round INV"))
    ROWS.append(("%d:16:16:0:1:PAIR" % L, "16-bit scaled-round PAIR"))
for L, L1 in ((20, 10), (21, 10), (22, 11), (24, 12)):
    ROWS.append(("%d:16:16:0:0:FWD:%d" % (L, L1), "16-bit scaled-trunc FWD, 2-D scheme 2^%d x 2^%d" % (L1, L - L1)))
ROWS.append(("20:16:16:0:0:INV:10", "16-bit scaled-trunc INV, 2-D scheme 2^10 x 2^10"))
ROWS.append(("21:16:16:0:0:INV:10", "16-bit scaled-trunc INV, 2-D scheme 2^10 x 2^11 (two launches, round 5)"))
ROWS.append(("20:16:16:0:0:PAIR:10", "16-bit scaled-trunc PAIR, 2-D scheme 2^10 x 2^10"))
ROWS.append(("21:16:16:0:0:PAIR:10", "16-bit scaled-trunc PAIR, 2-D scheme 2^10 x 2^11 (four launches, round 5)"))
ROWS.append(("20:16:16:1:0:FWD:10", "16-bit unscaled FWD, 2-D scheme 2^10 x 2^10 (36-bit results)"))
for L in (7, 10, 11, 12, 14, 16, 17, 18, 20):
    ROWS.append(("%d:16:16:0:0:INV" % L, "16-bit scaled-trunc INV"))
    ROWS.append(("%d:16:16:0:0:PAIR" % L, "16-bit scaled-trunc PAIR"))
for L in (7, 10, 11, 12, 14, 16):
    ROWS.append(("%d:16:16:1" % L, "16-bit unscaled FWD"))
for L in (7, 10):
    ROWS.append(("%d:16:16:1:0:INV" % L, "16-bit unscaled INV"))
ROWS.append(("7:16:16:1:0:PAIR", "16-bit unscaled PAIR"))
ROWS.append(("16:24:24:1", "24-bit unscaled FWD (C3)"))
ROWS.append(("16:24:16:1", "24-bit data / 16-bit twiddle unscaled FWD"))
for L in (11, 12, 13, 14, 15):
    ROWS.append(("%d:24:24:1" % L, "24-bit unscaled FWD"))
ROWS.append(("10:24:24:1", "24-bit unscaled FWD"))
ROWS.append(("7:24:24:1", "24-bit unscaled FWD"))
ROWS.append(("10:12:16:0", "12-bit scaled FWD (narrow data on the packed kernels)"))
ROWS.append(("10:12:16:0:0:PAIR", "12-bit scaled PAIR"))
ROWS.append(("12:14:16:0", "14-bit scaled FWD"))
ROWS.append(("12:14:16:0:0:PAIR", "14-bit scaled PAIR"))
ROWS.append(("16:12:16:0", "12-bit scaled FWD"))
ROWS.append(("20:12:16:0", "12-bit scaled FWD"))
ROWS.append(("10:18:18:0", "18-bit scaled FWD"))
for L in (14, 16, 20):
    ROWS.append(("%d:16:16:0:1" % L, "16-bit scaled-round FWD"))
ROWS.append(("16:16:16:0:1:INV", "16-bit scaled-round INV"))
ROWS.append(("20:16:16:0:1:INV", "16-bit scaled-round INV"))
ROWS.append(("14:16:16:0:1:PAIR", "16-bit scaled-round PAIR"))
ROWS.append(("13:16:16:0:1", "16-bit scaled-round FWD"))
ROWS.append(("13:16:16:0:1:INV", "16-bit scaled-round INV"))
ROWS.append(("14:16:16:0:1:INV", "16-bit scaled-round INV"))
ROWS.append(("13:16:16:0:1:PAIR", "16-bit scaled-round PAIR"))
ROWS.append(("10:18:18:0:0:INV", "18-bit scaled INV"))
ROWS.append(("12:14:16:0:1", "14-bit scaled-round FWD"))
ROWS.append(("12:14:16:0:1:PAIR", "14-bit scaled-round PAIR"))
ROWS.append(("10:12:16:0:1", "12-bit scaled-round FWD"))
ROWS.append(("12:18:18:0:1", "18-bit scaled-round FWD"))
ROWS.append(("10:32:24:0", "32-bit scaled FWD"))
ROWS.append(("12:32:24:0", "32-bit scaled FWD"))
ROWS.append(("10:12:16:0:0:INV", "12-bit scaled INV"))
ROWS.append(("12:16:16:1:0:INV", "16-bit unscaled INV"))
ROWS.append(("14:16:16:1:0:INV", "16-bit unscaled INV"))
ROWS.append(("16:16:16:1:0:INV", "16-bit unscaled INV"))
ROWS.append(("16:32:24:0", "32-bit scaled FWD"))
ROWS.append(("19:16:16:0", "16-bit scaled-trunc FWD"))
# round 4
ROWS.append(("15:16:16:0", "16-bit scaled-trunc FWD"))
for L in (13, 15):
    ROWS.append(("%d:16:16:0:0:INV" % L, "16-bit scaled-trunc INV"))
    ROWS.append(("%d:16:16:0:0:PAIR" % L, "16-bit scaled-trunc PAIR"))
for L in (17, 18, 19):
    ROWS.append(("%d:16:16:0:1" % L, "16-bit scaled-round FWD"))
ROWS.append(("18:16:16:0:1:INV", "16-bit scaled-round INV"))
ROWS.append(("23:16:16:0:0:FWD:10", "16-bit scaled-trunc FWD, 2-D scheme 2^10 x 2^13"))
ROWS.append(("22:16:16:0:0:FWD:10", "16-bit scaled-trunc FWD, 2-D scheme 2^10 x 2^12"))
ROWS.append(("22:16:16:0:0:INV:10", "16-bit scaled-trunc INV, 2-D scheme 2^10 x 2^12 (three launches, round 5)"))
ROWS.append(("22:16:16:0:0:INV:11", "16-bit scaled-trunc INV, 2-D scheme 2^11 x 2^11 (two launches, round 5)"))
ROWS.append(("22:16:16:0:0:PAIR:11", "16-bit scaled-trunc PAIR, 2-D scheme 2^11 x 2^11 (four launches, round 5)"))
ROWS.append(("23:16:16:0:0:INV:10", "16-bit scaled-trunc INV, 2-D scheme 2^10 x 2^13 (three launches, round 5)"))
for L in (13, 14, 16):
    ROWS.append(("%d:24:24:1:0:INV" % L, "24-bit unscaled INV (40-bit results)"))
ROWS.append(("16:24:16:1:0:INV", "24-bit data / 16-bit twiddle unscaled INV"))
ROWS.append(("19:16:16:0:0:INV", "16-bit scaled-trunc INV"))
ROWS.append(("10:32:16:1", "32-bit unscaled FWD (42-bit results)"))
ROWS.append(("7:32:16:1", "32-bit unscaled FWD (39-bit results)"))
ROWS.append(("10:32:24:1", "32-bit unscaled FWD, 24-bit twiddles"))
ROWS.append(("10:26:16:1:0:INV", "26-bit unscaled INV (the second half of a 16-bit unscaled pair)"))
ROWS.append(("10:34:24:1:0:INV", "34-bit unscaled INV, 24-bit twiddles (three-dword products)"))
ROWS.append(("10:16:16:1:0:PAIR", "16-bit unscaled PAIR (36-bit results)"))
ROWS.append(("10:24:24:1:0:PAIR", "24-bit unscaled PAIR (44-bit results)"))
ROWS.append(("7:32:16:1:0:PAIR", "32-bit unscaled PAIR (46-bit results)"))
ROWS.append(("10:40:16:0", "40-bit scaled FWD"))
ROWS.append(("12:32:16:1", "32-bit unscaled FWD (44-bit results)"))
ROWS.append(("11:32:16:1", "32-bit unscaled FWD (43-bit results)"))
ROWS.append(("12:28:16:1:0:INV", "28-bit unscaled INV (the second half of a 16-bit unscaled pair)"))
ROWS.append(("12:16:16:1:0:PAIR", "16-bit unscaled PAIR (40-bit results)"))
ROWS.append(("16:24:24:1:0:INV", "24-bit unscaled INV (40-bit results)"))
ROWS.append(("10:18:16:0:0:PAIR", "18-bit scaled PAIR (32-bit words: forward + inverse sub-plans, round 4)"))
ROWS.append(("12:24:24:0:1:PAIR", "24-bit scaled-round PAIR (32-bit words)"))
ROWS.append(("14:18:16:0:0:PAIR", "18-bit scaled PAIR (32-bit words, two passes each way)"))
for _l in (13, 14, 16):
    ROWS.append(("%d:32:16:1" % _l, "32-bit unscaled FWD (%d-bit results; both passes on 64-bit words, round 5)" % (32 + _l)))
    ROWS.append(("%d:32:16:1:0:INV" % _l, "32-bit unscaled INV (%d-bit results; round 5)" % (32 + _l)))
ROWS.append(("16:28:16:1", "28-bit unscaled FWD (44-bit results; round 5)"))
ROWS.append(("16:24:12:1", "24-bit unscaled FWD, 12-bit twiddles (round 5: outside the int32-first-pass class)"))
ROWS.append(("10:56:16:1", "56-bit unscaled FWD (66-bit results, 16-byte containers)"))
ROWS.append(("10:58:12:1:0:INV", "58-bit unscaled INV, 12-bit twiddles (68-bit results)"))

# round 5: the long frames outside 16-bit scaled data (csrc/intfft_widelong.hip, csrc/intfft_bigwlong.hip)
for L in (17, 18, 19, 20):
    ROWS.append(("%d:16:16:1" % L, "16-bit unscaled FWD (%d-bit results; three launches, round 5)" % (16 + L)))
ROWS.append(("17:20:16:1", "20-bit unscaled FWD (37-bit results; round 5)"))
for L in (17, 20):
    ROWS.append(("%d:16:16:1:0:INV" % L, "16-bit unscaled INV (%d-bit results; gather pass, block pass, 64-bit post-pass, round 5)" % (16 + L)))
ROWS.append(("17:24:24:1", "24-bit unscaled FWD (41-bit results; 64-bit first pass on the blocks, round 5)"))
ROWS.append(("20:24:16:1", "24-bit data / 16-bit twiddle unscaled FWD (44-bit results; round 5)"))
ROWS.append(("17:24:24:1:0:INV", "24-bit unscaled INV (41-bit results; every stage on 64-bit words, round 5)"))
for L in (17, 20):
    ROWS.append(("%d:18:18:0" % L, "18-bit scaled FWD (int32 words, pre-pass + two passes, round 5)"))
    ROWS.append(("%d:18:18:0:0:INV" % L, "18-bit scaled INV (two passes + post-pass, round 5)"))
ROWS.append(("17:24:24:0:1", "24-bit scaled-round FWD (round 5)"))
ROWS.append(("18:32:24:0", "32-bit scaled FWD (round 5)"))
ROWS.append(("18:32:24:0:0:INV", "32-bit scaled INV (round 5)"))
ROWS.append(("17:12:16:1", "12-bit unscaled FWD (29-bit results; round 5)"))
ROWS.append(("20:12:16:1", "12-bit unscaled FWD (32-bit results; round 5)"))
ROWS.append(("17:12:16:1:0:INV", "12-bit unscaled INV (round 5)"))
ROWS.append(("18:18:18:0:0:PAIR", "18-bit scaled PAIR (forward + inverse sub-plans on the long-frame kernels, round 5)"))

NATIVE = [("7:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out (native int_fftNk beats)"),
          ("12:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out"),
          ("16:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out"),
          ("20:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out"),
          ("20:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out"),
          ("7:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out (native int_ifftNk beats)"),
          ("12:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out"),
          ("16:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out"),
          ("13:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out (one pass, round 4)"),
          ("14:16:16:0", ("HALVES", "BITREV"), "16-bit scaled-trunc FWD, HALVES in / BITREV out (one pass, round 4)"),
          ("13:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out (one pass, round 4)"),
          ("14:16:16:0:0:INV", ("BITREV", "HALVES"), "16-bit scaled-trunc INV, BITREV in / HALVES out (one pass, round 4)"),
          ("14:16:16:0:1", ("HALVES", "BITREV"), "16-bit scaled-round FWD, HALVES in / BITREV out (one pass, round 4)"),
          ("10:16:16:1", ("HALVES", "BITREV"), "16-bit unscaled FWD, HALVES in / BITREV out (round 4)"),
          ("7:16:16:1", ("HALVES", "BITREV"), "16-bit unscaled FWD, HALVES in / BITREV out (round 4)"),
          ("10:16:16:1:0:INV", ("BITREV", "HALVES"), "16-bit unscaled INV, BITREV in / HALVES out (round 4)"),
          ("7:16:16:1:0:INV", ("BITREV", "HALVES"), "16-bit unscaled INV, BITREV in / HALVES out (round 4)"),
          ("10:18:16:0", ("HALVES", "BITREV"), "18-bit scaled FWD, HALVES in / BITREV out (round 4)"),
          ("10:18:16:0:0:INV", ("BITREV", "HALVES"), "18-bit scaled INV, BITREV in / HALVES out (round 4)"),
          ("7:24:24:1", ("HALVES", "BITREV"), "24-bit unscaled FWD, HALVES in / BITREV out (round 4)"),
          ("16:24:24:1", ("HALVES", "BITREV"), "24-bit unscaled FWD (40-bit results), HALVES in / BITREV out (round 5)"),
          ("16:24:24:1:0:INV", ("BITREV", "HALVES"), "24-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("13:24:24:1", ("HALVES", "BITREV"), "24-bit unscaled FWD, HALVES in / BITREV out (round 5)"),
          ("16:32:16:1", ("HALVES", "BITREV"), "32-bit unscaled FWD (48-bit results), HALVES in / BITREV out (round 5)"),
          ("16:32:16:1:0:INV", ("BITREV", "HALVES"), "32-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("14:16:16:1", ("HALVES", "BITREV"), "16-bit unscaled FWD (30-bit results), HALVES in / BITREV out (round 5)"),
          ("14:16:16:1:0:INV", ("BITREV", "HALVES"), "16-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("16:18:16:0", ("HALVES", "BITREV"), "18-bit scaled FWD, HALVES in / BITREV out (round 5)"),
          ("16:32:24:0:0:INV", ("BITREV", "HALVES"), "32-bit scaled INV, BITREV in / HALVES out (round 5)"),
          ("10:32:16:1", ("HALVES", "BITREV"), "32-bit unscaled FWD (42-bit results), HALVES in / BITREV out (round 5)"),
          ("10:32:16:1:0:INV", ("BITREV", "HALVES"), "32-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("10:48:24:0", ("HALVES", "BITREV"), "48-bit scaled FWD, 24-bit twiddles, HALVES in / BITREV out (round 5)"),
          ("7:32:16:1", ("HALVES", "BITREV"), "32-bit unscaled FWD (39-bit results), HALVES in / BITREV out (round 5)"),
          ("7:32:16:1:0:INV", ("BITREV", "HALVES"), "32-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("12:32:16:1", ("HALVES", "BITREV"), "32-bit unscaled FWD (44-bit results), HALVES in / BITREV out (round 5)"),
          ("12:32:16:1:0:INV", ("BITREV", "HALVES"), "32-bit unscaled INV, BITREV in / HALVES out (round 5)"),
          ("11:40:16:0", ("HALVES", "BITREV"), "40-bit scaled FWD, HALVES in / BITREV out (round 5)"),
          ("17:16:16:1", ("HALVES", "BITREV"), "16-bit unscaled FWD (33-bit results), HALVES in / BITREV out (long frames, round 5)"),
          ("20:16:16:1", ("HALVES", "BITREV"), "16-bit unscaled FWD (36-bit results), HALVES in / BITREV out (long frames, round 5)"),
          ("17:24:24:1", ("HALVES", "BITREV"), "24-bit unscaled FWD (41-bit results), HALVES in / BITREV out (long frames, round 5)"),
          ("17:16:16:1:0:INV", ("BITREV", "HALVES"), "16-bit unscaled INV, BITREV in / HALVES out (long frames, round 5)"),
          ("17:18:18:0", ("HALVES", "BITREV"), "18-bit scaled FWD, HALVES in / BITREV out (long frames, round 5)"),
          ("17:18:18:0:0:INV", ("BITREV", "HALVES"), "18-bit scaled INV, BITREV in / HALVES out (long frames, round 5)")]

# round 6: BITREV_LANES at one end of a single core and USE_FLY = 0 (full ad-hoc specs "L:DW:TW:FMT:RND:DIR:L1:IN:OUT[:USE_FLY]")
ROUND6 = [("10:16:16:0:0:FWD:0:NATURAL:BITREV_LANES", "16-bit scaled-trunc FWD, BITREV_LANES out (store map, round 6)"),
          ("10:16:16:0:0:INV:0:BITREV_LANES:NATURAL", "16-bit scaled-trunc INV, BITREV_LANES in (load map, round 6)"),
          ("10:16:16:0:1:FWD:0:HALVES:BITREV_LANES", "16-bit scaled-round FWD, HALVES in / BITREV_LANES out (round 6)"),
          ("7:16:16:0:0:FWD:0:NATURAL:BITREV_LANES", "16-bit scaled-trunc FWD, BITREV_LANES out (round 6)"),
          ("12:16:16:0:0:FWD:0:NATURAL:BITREV_LANES", "16-bit scaled-trunc FWD, BITREV_LANES out (round 6)"),
          ("12:16:16:0:0:INV:0:BITREV_LANES:NATURAL", "16-bit scaled-trunc INV, BITREV_LANES in (round 6)"),
          ("14:16:16:0:0:FWD:0:NATURAL:BITREV_LANES", "16-bit scaled-trunc FWD, BITREV_LANES out (round 6)"),
          ("16:16:16:0:0:FWD:0:HALVES:BITREV_LANES", "16-bit scaled-trunc FWD, HALVES in / BITREV_LANES out (round 6)"),
          ("20:16:16:0:0:FWD:0:NATURAL:BITREV_LANES", "16-bit scaled-trunc FWD, BITREV_LANES out (round 6)"),
          ("10:24:24:1:0:FWD:0:NATURAL:BITREV_LANES", "24-bit unscaled FWD, BITREV_LANES out (round 6)"),
          ("16:24:24:1:0:FWD:0:NATURAL:BITREV_LANES", "24-bit unscaled FWD (C3's plan), BITREV_LANES out (round 6)"),
          ("10:16:16:0:0:FWD:0:NATURAL:BITREV:0", "16-bit scaled FWD, USE_FLY = 0, natural in / BITREV out (no data movement, round 6)"),
          ("10:16:16:0:0:FWD:0:NATURAL:NATURAL:0", "16-bit scaled FWD, USE_FLY = 0, natural in / natural out (round 6)"),
          ("16:24:24:1:0:INV:0:BITREV:NATURAL:0", "24-bit unscaled INV, USE_FLY = 0 (round 6)"),
          ("12:16:16:0:0:PAIR:0:HALVES:NATURAL:0", "16-bit scaled PAIR, USE_FLY = 0, HALVES in (round 6)")]

if __name__ == "__main__":
    print("Every row: one call on 256 MiB of input, 10 timed steps after a clock ramp.  At the multi-pass "
          "lengths that is ONE scratch chunk per call, so the two-stream chunk alternation of N = 2^19 / 2^20, the 24-bit class and the tiled "
          "2-D plans does not show here: BASELINE's batches are `python bench.py --config C3 | C4 | C5` (profiles/rNN_other_configs_bench.jsonl).\n")
    print("| N | mode | kernel | passes | Gsample/s | B/sample | GB/s | frac of 8 TB/s | parity prefix |")
    print("|---|---|---|---|---|---|---|---|---|")
    seen = set()
    for spec, label in ROWS:
        if spec in seen:
            continue
        seen.add(spec)
        B.adhoc(spec)
        r = B.run(spec, steps=10)
        bps = r["GB/s"] / r["Gsample/s"]
        print("| 2^%d | %s | `%s` | %d | %.0f | %.0f | %.0f | %.2f | %s |" % (
            r["log2n"], label, r["kernel"], r["passes"], r["Gsample/s"], bps, r["GB/s"], r["roofline_frac"],
            "ok" if r["parity_prefix_ok"] else "MISMATCH"), flush=True)
    for spec, orders, label in NATIVE:
        B.adhoc(spec)
        B.ORDERS[spec] = orders
        r = B.run(spec, steps=10)
        bps = r["GB/s"] / r["Gsample/s"]
        print("| 2^%d | %s | `%s` | %d | %.0f | %.0f | %.0f | %.2f | %s |" % (
            r["log2n"], label, r["kernel"], r["passes"], r["Gsample/s"], bps, r["GB/s"], r["roofline_frac"],
            "ok" if r["parity_prefix_ok"] else "MISMATCH"), flush=True)
    for spec, label in ROUND6:
        B.adhoc(spec)
        r = B.run(spec, steps=10)
        bps = r["GB/s"] / r["Gsample/s"]
        print("| 2^%d | %s | `%s` | %d | %.0f | %.0f | %.0f | %.2f | %s |" % (
            r["log2n"], label, r["kernel"], r["passes"], r["Gsample/s"], bps, r["GB/s"], r["roofline_frac"],
            "ok" if r["parity_prefix_ok"] else "MISMATCH"), flush=True)
