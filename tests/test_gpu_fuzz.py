"""A bounded, seeded parity fuzz inside the driver-run GPU suite (the open-ended soaks are tools/fuzz_soak.py / tools/fuzz_wide.py).

~1200 random elaboratable configurations -- NFFT 3 .. 20 weighted to short frames, DATA_WIDTH 4 .. 64, TWDL_WIDTH 8 .. 26, the three
modes, the three directions, both XSERIES, every I/O order pair, ragged batches, now and then the 2-D scheme with a random split --
whatever kernel the planner picks must equal the C oracle bit for bit ("three modes side by side ... batch index
independence"); results beyond 64 bits (16-byte containers) are checked against the Python twin on short frames.  The generator is a
pure function of the chunk's seed, so a kernel family added late in a round is covered by construction the next time the suite runs.
A failure prints the seed of the chunk, the index inside it and the generics."""
import time

import numpy as np
import pytest

from tests.helpers import edge_frames, uniform_frames

pytestmark = pytest.mark.gpu

NP = {2: np.int16, 4: np.int32, 8: np.int64}
ORDERS = ["NATURAL", "BITREV", "HALVES", "BITREV_LANES"]
# weighted to short frames: the long ones cost the oracle seconds, and tests/test_gpu_parity.py walks every length on fixed cases
LOG2N = [3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 16, 16, 17, 18, 19, 20]
CHUNKS = 24  # 1200 configurations (round 6; 8 chunks up to round 5): ~30 s on the GPU box
PER_CHUNK = 50


def draw(rng):
    """One candidate configuration (may be non-elaboratable: the caller asks the oracle's validator)."""
    log2n = int(rng.choice(LOG2N))
    fmt = int(rng.integers(0, 2))
    rnd = 0 if fmt else int(rng.integers(0, 2))
    dw = int(rng.choice([16, 16, 16, 12, 14, 18, 24, 24, 32, int(rng.integers(4, 65))]))
    tw = int(rng.choice([16, 16, 24, int(rng.integers(8, 27))]))
    new = bool(rng.integers(0, 2))
    d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
    if rng.random() < 0.4:
        in_o, out_o = ORDERS[int(rng.integers(0, 4))], ORDERS[int(rng.integers(0, 4))]
    else:
        in_o = out_o = "NATURAL"
    l1 = 0
    if 6 <= log2n <= 16 and rng.random() < 0.12:
        l1 = int(rng.integers(3, log2n - 2))
    elif log2n == 20 and rng.random() < 0.4:
        l1 = 10
    if log2n >= 19:
        batch = int(rng.integers(1, 3))
    elif log2n >= 17:
        batch = int(rng.integers(1, 4))
    elif log2n >= 12:
        batch = int(rng.integers(1, 40))
    else:
        batch = int(rng.integers(1, 300))
    bits = dw if rng.random() < 0.5 else max(2, dw - 1)
    use_fly = 0 if (not l1 and rng.random() < 0.06) else 1  # the bypass mux (round 6: a plan of its own)
    return dict(log2n=log2n, dw=dw, tw=tw, fmt=fmt, rnd=rnd, new=new, d=d, in_o=in_o, out_o=out_o, l1=l1, batch=batch, bits=bits, use_fly=use_fly,
                dseed=int(rng.integers(1, 1 << 30)), edges=int(rng.integers(1, 7)) if rng.random() < 0.3 else 0)


def configurations(seed, count):
    """`count` elaboratable configurations whose results fit 64 bits, a pure function of `seed`."""
    from oracle import oracle_c as C

    dirs = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        c = draw(rng)
        p = C.make_params(c["log2n"], c["dw"], c["tw"], c["fmt"], c["rnd"], c["new"], c["use_fly"])
        if c["l1"]:
            if "BITREV_LANES" in (c["in_o"], c["out_o"]) or C.lib().orc_validate_2d(p, c["l1"], dirs[c["d"]]) != 0:
                continue
        elif C.lib().orc_validate(p, dirs[c["d"]]) != 0:
            continue
        if c["dw"] + c["fmt"] * c["log2n"] * (2 if c["d"] == "PAIR" else 1) > 64:
            continue  # 16-byte containers: test_fuzz_results_beyond_64_bits
        out.append(c)
    return out


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_fuzz_against_the_oracle(chunk):
    import torch

    from intfftk_amd import IntFFTCore
    from oracle import oracle_c as C

    dirs = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
    ords = {"NATURAL": C.NATURAL, "BITREV": C.BITREV, "HALVES": C.HALVES, "BITREV_LANES": C.BITREV_LANES}
    seed = 0x5EED0500 + chunk
    t0 = time.time()
    bad = []
    kernels = set()
    for i, c in enumerate(configurations(seed, PER_CHUNK)):
        n = 1 << c["log2n"]
        x = uniform_frames(c["batch"], n, c["bits"], c["dseed"])
        if c["edges"]:
            x = np.concatenate([x, edge_frames(n, c["dw"])[: c["edges"]]])
        core = IntFFTCore(c["log2n"], c["dw"], c["tw"], c["fmt"], c["rnd"], "NEW" if c["new"] else "OLD", c["d"], c["in_o"], c["out_o"],
                          c["use_fly"], NFFT1=c["l1"])
        y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
        torch.cuda.synchronize()
        got = y.cpu().numpy().astype(np.int64)
        name = core.info["kernel_name"]
        kernels.add(name)
        core.close()
        p = C.make_params(c["log2n"], c["dw"], c["tw"], c["fmt"], c["rnd"], c["new"], c["use_fly"])
        if c["l1"]:
            want = C.execute_2d(x, p, c["l1"], dirs[c["d"]], ords[c["in_o"]], ords[c["out_o"]], form=1)
        else:
            want = C.execute(x, p, dirs[c["d"]], ords[c["in_o"]], ords[c["out_o"]], form=1)
        if got.shape != want.shape or not np.array_equal(got, want):
            bad.append((hex(seed), i, name, c))
    print("fuzz chunk %d (seed %#x): %d configurations, %d kernel families, %.1f s" % (chunk, seed, PER_CHUNK, len(kernels), time.time() - t0))
    assert not bad, bad


def test_fuzz_results_beyond_64_bits():
    """The 64-bit boundary and the 16-byte containers (k_pass<__int128>) against the Python twin (big integers): short frames only."""
    import torch

    from intfftk_amd import IntFFTCore
    from intfftk_amd.engine import wide_to_int
    from oracle import oracle_py as P

    dirs = {"FWD": P.FWD, "INV": P.INV, "PAIR": P.PAIR}
    seed = 0x5EED05FF
    rng = np.random.default_rng(seed)
    done, bad, wide = 0, [], 0
    while done < 24:
        log2n = int(rng.integers(3, 8))
        dw = int(rng.integers(30, 65))
        tw = int(rng.integers(8, 27))
        new = bool(rng.integers(0, 2))
        d = ["FWD", "INV", "PAIR"][int(rng.integers(0, 3))]
        fmt = 1 if rng.random() < 0.8 else 0
        rnd = 0 if fmt else int(rng.integers(0, 2))
        ob = dw + fmt * log2n * (2 if d == "PAIR" else 1)
        if ob < 60 or ob > 96:
            continue
        try:
            core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW" if new else "OLD", d)
        except Exception:
            continue  # not elaboratable (the planner's accept set is the oracle's: tests/test_capi_cpu.py)
        n = 1 << log2n
        x = np.concatenate([uniform_frames(2, n, dw, int(rng.integers(1, 1 << 30))), edge_frames(n, dw)[[1, 4]]])
        y = core(torch.from_numpy(np.ascontiguousarray(x.astype(NP[core.in_container]))).cuda())
        torch.cuda.synchronize()
        info = dict(core.info)
        core.close()
        y = y.cpu().numpy()
        got = wide_to_int(y) if info["out_container"] == 16 else y.astype(object)
        wide += info["out_container"] == 16
        for f in range(x.shape[0]):
            want = P.execute([(int(a), int(b)) for a, b in x[f]], log2n, dw, tw, fmt, rnd, new, dirs[d])
            if any((int(got[f, m, 0]), int(got[f, m, 1])) != w for m, w in enumerate(want)):
                bad.append((hex(seed), done, info["kernel_name"], (log2n, dw, tw, fmt, rnd, new, d), f))
                break
        done += 1
    assert not bad, bad
    assert wide >= 4  # the draw really reaches the 16-byte containers
