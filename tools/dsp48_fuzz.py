#!/usr/bin/env python3
"""Fuzz of the DSP48-primitive-level structural twin (oracle/dsp48_twin.py) against the "slice of P" oracles
(oracle/oracle_py.py and, where the result fits 64 bits, oracle/intfft_oracle.c).  TEST INFRASTRUCTURE, CPU only.

  python tools/dsp48_fuzz.py [--per-case N] [--procs P] [--seed S]   ->  one line per (entity, widths, XSER) + a summary

Covers: every exact multiplier (mlt*), every int_cmult_dsp48 regime x XSER over all its elaboratable widths,
int_addsub_dsp48 for DSPW = 2 .. 95 x XSER, row_twiddle_tay for STAGE 11 .. 19 x AWD x XSER x USE_MLT, and the two butterflies
(int_dif2_fly / int_dit2_fly: adder + rounding / negation processes + multiplier) over random widths, modes and stages.
Operand mix per case: uniform, corner values (min / max / -1 / 0 / +-1) and "carry-edge" values whose low 17 / 34 / 48
bits are all ones or all zeros (the places where a wrong cascade or carry chain would show).
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import dsp48_twin as tw  # noqa: E402
from oracle import oracle_py as op  # noqa: E402


def operand(rng: random.Random, w: int) -> int:
    """A signed w-bit operand from the mix described above."""
    lo, hi = -(1 << (w - 1)), (1 << (w - 1)) - 1
    k = rng.random()
    if k < 0.70:
        return rng.randint(lo, hi)
    if k < 0.80:
        return rng.choice((lo, hi, -1, 0, 1, lo + 1, hi - 1))
    v = rng.randint(lo, hi)
    cut = rng.choice((17, 18, 34, 35, 48))
    if cut < w:
        m = (1 << cut) - 1
        v = (v | m) if rng.random() < 0.5 else (v & ~m)
    return max(lo, min(hi, v))


def cmult_cases(new: bool):
    """Every (w, t) int_cmult_dsp48 elaborates for this XSER, labelled with its regime."""
    for t in range(8, 28):
        for w in range(8, 79):
            r = op.cmult_regime(w, t, new)
            if r is not None:
                yield r, w, t


def run_cmult(new: bool, w: int, t: int, n: int, seed: int):
    rng = random.Random(seed)
    xser = "NEW" if new else "OLD"
    bad = 0
    for _ in range(n):
        dr, di, wr, wi = operand(rng, w), operand(rng, w), operand(rng, t), operand(rng, t)
        got = tw.int_cmult_dsp48(tw.vec(dr, w), tw.vec(di, w), tw.vec(wr, t), tw.vec(wi, t), w, t, xser)
        want = op.cmult(dr, di, wr, wi, w, t, new)
        if got is None or (tw.signed(got[0], w), tw.signed(got[1], w)) != want:
            bad += 1
            if bad <= 3:
                print("MISMATCH cmult", xser, w, t, (dr, di, wr, wi), got, want, flush=True)
    return bad


def run_mlt(name: str, n: int, seed: int):
    rng = random.Random(seed)
    fn, aw, bw = {
        "mlt42x18_dsp48e1": (tw.mlt42x18_dsp48e1, 42, 18), "mlt44x18_dsp48e2": (tw.mlt44x18_dsp48e2, 44, 18),
        "mlt35x25_dsp48e1": (tw.mlt35x25_dsp48e1, 35, 25), "mlt35x27_dsp48e2": (tw.mlt35x27_dsp48e2, 35, 27),
        "mlt59x18_dsp48e1": (tw.mlt59x18_dsp48e1, 59, 18), "mlt61x18_dsp48e2": (tw.mlt61x18_dsp48e2, 61, 18),
        "mlt52x25_dsp48e1": (tw.mlt52x25_dsp48e1, 52, 25), "mlt52x27_dsp48e2": (tw.mlt52x27_dsp48e2, 52, 27),
    }[name]
    bad = 0
    for _ in range(n):
        a, b = operand(rng, aw), operand(rng, bw)
        got = tw.signed(fn(tw.vec(a, aw), tw.vec(b, bw)), aw + bw)
        if got != a * b:
            bad += 1
            if bad <= 3:
                print("MISMATCH", name, a, b, got, a * b, flush=True)
    return bad


def run_addsub(new: bool, dspw: int, n: int, seed: int):
    rng = random.Random(seed)
    xser = "NEW" if new else "OLD"
    bad = 0
    for _ in range(n):
        v = [operand(rng, dspw) for _ in range(4)]
        got = tw.int_addsub_dsp48(*(tw.vec(x, dspw) for x in v), dspw, xser)
        want = (v[0] + v[2], v[1] + v[3], v[0] - v[2], v[1] - v[3])
        if tuple(tw.signed(g, dspw + 1) for g in got) != want:
            bad += 1
            if bad <= 3:
                print("MISMATCH addsub", xser, dspw, v, got, want, flush=True)
    return bad


def run_taylor(new: bool, stage: int, t: int, n: int, seed: int):
    """row_twiddle_tay fed the way rom_twiddle_int.vhd:171-244 feeds it, against oracle_py.twiddles(stage)."""
    rng = random.Random(seed)
    xser = "NEW" if new else "OLD"
    rom = op._rom(9, t)
    want = op.twiddles(stage, t, new)
    bad = 0
    for i in range(n):
        cnt = rng.randrange(1 << stage) if n < (1 << stage) else i
        div, addr = (cnt >> (stage - 1)) & 1, cnt & ((1 << (stage - 1)) - 1)
        re, im = rom[addr >> (stage - 10)]
        if div:
            re, im = im, -re
        ww_rom = tw.vec(re, t) | (tw.vec(im, t) << t)
        count = addr & ((1 << (stage - 10)) - 1)
        for use_mlt in (False, True):
            g = tw.row_twiddle_tay(ww_rom, count, t, xser, stage - 11, use_mlt)
            if (tw.signed(g[0], t), tw.signed(g[1], t)) != want[cnt]:
                bad += 1
                if bad <= 3:
                    print("MISMATCH taylor", xser, stage, t, cnt, use_mlt, g, want[cnt], flush=True)
    return bad


def run_fly(new: bool, kind: str, n: int, seed: int):
    """int_dif2_fly / int_dit2_fly wired through int_addsub_dsp48 + pr_rnd / pr_inv + int_cmult_dsp48, against oracle_py.dif_fly / dit_fly."""
    rng = random.Random(seed)
    xser = "NEW" if new else "OLD"
    bad = 0
    for _ in range(n):
        dtw, tfw = rng.randint(8, 60), rng.choice((8, 12, 16, 18, 19, 24, 25))
        scale = rng.randint(0, 1)
        rnd = rng.randint(0, 1) if scale else 0
        stage, odd = rng.choice((0, 1, 2, 3, 10, 11, 15)), rng.randint(0, 1)
        if stage > 1 and op.cmult_regime(dtw + 1 - scale if kind == "dif" else dtw, tfw, new) is None:
            continue
        v = [operand(rng, dtw) for _ in range(4)]
        ww = (operand(rng, tfw), operand(rng, tfw))
        f = tw.int_dif2_fly if kind == "dif" else tw.int_dit2_fly
        g = f(*(tw.vec(x, dtw) for x in v), tw.vec(ww[0], tfw), tw.vec(ww[1], tfw), stage=stage, scale=scale, dtw=dtw, tfw=tfw,
              rndmode=rnd, xser=xser, dt_sw=odd)
        wo = dtw - scale + 1
        o = (op.dif_fly if kind == "dif" else op.dit_fly)((v[0], v[1]), (v[2], v[3]), ww, stage, dtw, tfw, scale, rnd, odd, new)
        want = tuple(op.sgn(x, wo) for x in (o[0][0], o[0][1], o[1][0], o[1][1]))
        if g is None or tuple(tw.signed(x, wo) for x in g) != want:
            bad += 1
            if bad <= 3:
                print("MISMATCH fly", kind, xser, dtw, tfw, scale, rnd, stage, odd, v, ww, g, want, flush=True)
    return bad


def job(args):
    kind, key, n, seed = args
    if kind == "fly":
        return kind, key, n, run_fly(*key, n, seed)
    if kind == "cmult":
        return kind, key, n, run_cmult(*key, n, seed)
    if kind == "mlt":
        return kind, key, n, run_mlt(key, n, seed)
    if kind == "addsub":
        return kind, key, n, run_addsub(*key, n, seed)
    return kind, key, n, run_taylor(*key, n, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--per-case", type=int, default=2000)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--seed", type=int, default=20260930)
    a = ap.parse_args()
    jobs = []
    for name in ("mlt42x18_dsp48e1", "mlt44x18_dsp48e2", "mlt35x25_dsp48e1", "mlt35x27_dsp48e2", "mlt59x18_dsp48e1",
                 "mlt61x18_dsp48e2", "mlt52x25_dsp48e1", "mlt52x27_dsp48e2"):
        jobs.append(("mlt", name, a.per_case * 20))
    regimes = {}
    for new in (True, False):
        for r, w, t in cmult_cases(new):
            jobs.append(("cmult", (new, w, t), a.per_case))
            regimes[(new, w, t)] = r
        for dspw in range(2, 96):  # OX(DSPW downto 48) <= P2(DSPW-48 downto 0) needs DSPW - 48 <= 47
            jobs.append(("addsub", (new, dspw), a.per_case))
        for stage in range(11, 20):
            for t in (12, 16, 18, 19, 24, 25) + ((27,) if new else ()):
                jobs.append(("taylor", (new, stage, t), min(a.per_case, 1 << stage)))
        for kind in ("dif", "dit"):
            for _ in range(16):
                jobs.append(("fly", (new, kind), a.per_case * 4))
    jobs = [(k, key, n, a.seed + i) for i, (k, key, n) in enumerate(jobs)]
    t0 = time.time()
    if a.procs > 1:
        import multiprocessing as mp
        with mp.Pool(a.procs) as pool:
            res = pool.map(job, jobs, chunksize=4)
    else:
        res = [job(j) for j in jobs]
    tot = {}
    for kind, key, n, bad in res:
        label = kind if kind != "cmult" else "cmult:%s:%s" % (regimes[key], "NEW" if key[0] else "OLD")
        c = tot.setdefault(label, [0, 0, 0])
        c[0] += 1
        c[1] += n
        c[2] += bad
    print("%-26s %8s %12s %10s" % ("entity", "cases", "operand sets", "mismatches"))
    for label in sorted(tot):
        print("%-26s %8d %12d %10d" % (label, *tot[label]))
    nbad = sum(c[2] for c in tot.values())
    print("total: %d cases, %d operand sets, %d mismatches, %.0f s on %d processes, seed %d"
          % (len(jobs), sum(c[1] for c in tot.values()), nbad, time.time() - t0, a.procs, a.seed))
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
