"""The frame-queue form of the streaming host interface (include/intfft.h: intfft_stream_*): frames
arrive over time in chunks of 1 .. 64 with gaps (the software picture of the RTL's valid strobes, int_fftNk.vhd:23-37), results
leave in push order and are bit-equal to the oracle."""
import random
import threading
import time

import numpy as np
import pytest

from oracle import oracle_c as C
from tests.helpers import edge_frames, uniform_frames

pytestmark = pytest.mark.gpu

DIR = {"FWD": C.FWD, "INV": C.INV, "PAIR": C.PAIR}
NPDT = {2: np.int16, 4: np.int32, 8: np.int64}


def oracle(x, log2n, dw, tw, fmt, rnd, d):
    return C.execute(x, C.make_params(log2n, dw, tw, fmt, rnd, True), DIR[d], C.NATURAL, C.NATURAL, form=1)


@pytest.mark.parametrize("cfg,frames,slot_frames,n_slots", [
    ((10, 16, 16, 0, 0, "FWD"), 1500, 64, 3),      # the headline shape, many small slots: every slot is reused many times
    ((10, 16, 16, 0, 1, "PAIR"), 700, 100, 2),      # round mode pair, two slots (the minimum)
    ((8, 24, 16, 1, 0, "FWD"), 900, 37, 4),         # int32 in -> int32 out, odd slot size
    ((15, 16, 16, 0, 0, "FWD"), 40, 8, 3),          # a multi-pass plan (workspace owned by the stream object)
    ((12, 16, 16, 0, 0, "INV"), 300, 0, 0),         # library defaults
])
def test_producer_thread_with_random_gaps(cfg, frames, slot_frames, n_slots):
    """A producer thread pushes 1 .. 64-frame chunks with random pauses (and flushes now and then, as a source that goes quiet
    would); the consumer thread pulls whatever is ready.  Everything comes out, in order, bit-equal to the oracle."""
    from intfftk_amd import FrameStream, IntFFTCore

    log2n, dw, tw, fmt, rnd, d = cfg
    n = 1 << log2n
    core = IntFFTCore(log2n, dw, tw, fmt, rnd, "NEW", d)
    x = uniform_frames(frames, n, dw, 77)
    x[:8] = edge_frames(n, dw)
    xin = x.astype(NPDT[core.in_container])
    want = oracle(x, log2n, dw, tw, fmt, rnd, d)
    st = FrameStream(core, slot_frames, n_slots)
    err = []

    def producer():
        try:
            rng = random.Random(5)
            pos = 0
            while pos < frames:
                k = min(rng.randint(1, 64), frames - pos)
                while k:  # push never waits: when the ring is full it takes fewer frames and the consumer has to catch up
                    a = st.push(xin[pos:pos + k])
                    pos += a
                    k -= a
                    if k:
                        time.sleep(0.0005)
                r = rng.random()
                if r < 0.15:
                    time.sleep(rng.random() * 0.003)  # a gap between bursts
                elif r < 0.25:
                    st.flush()  # the source goes quiet: get the short chunk out
            st.flush()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=producer)
    t.start()
    parts, got_n = [], 0
    deadline = time.time() + 120
    while got_n < frames and time.time() < deadline and not err:
        y = st.pull(97, wait=False)
        if len(y):
            parts.append(y.copy())
            got_n += len(y)
        else:
            time.sleep(0.0002)
    t.join()
    assert not err, err
    assert got_n == frames
    got = np.concatenate(parts).astype(np.int64)
    assert np.array_equal(got, want)
    assert st.pending() == (0, 0)
    st.close()
    core.close()


def test_ring_full_flush_and_blocking_pull():
    """Single-threaded use: push stops accepting when every slot holds results that were not pulled; frames of a slot that was
    not submitted are invisible to pull until flush; a blocking pull returns the oldest slot."""
    from intfftk_amd import FrameStream, IntFFTCore

    core = IntFFTCore(9, 16, 16, 0, 0, "NEW", "FWD")
    x = uniform_frames(100, 512, 16, 3)
    xin = x.astype(np.int16)
    want = oracle(x, 9, 16, 16, 0, 0, "FWD")
    st = FrameStream(core, 10, 2)
    assert st.push(xin[:5]) == 5
    assert st.pending() == (5, 5)
    assert len(st.pull(100, wait=True)) == 0           # nothing submitted: a blocking pull does not wait for the filling slot
    assert st.push(xin[5:40]) == 15                    # two slots of ten: 20 frames in flight, the rest is refused
    assert st.pending() == (20, 0)
    assert st.push(xin[20:21]) == 0
    a = st.pull(4, wait=True)                          # part of the oldest slot
    assert np.array_equal(a.astype(np.int64), want[:4])
    assert st.push(xin[20:21]) == 0                    # that slot still holds six frames
    b = st.pull(100, wait=True)
    assert len(b) >= 6
    got = np.concatenate([a, b])
    while len(got) < 20:
        got = np.concatenate([got, st.pull(100, wait=True)])
    assert np.array_equal(got.astype(np.int64), want[:20])
    assert st.push(xin[20:27]) == 7
    st.flush()
    c = st.pull(100, wait=True)
    assert np.array_equal(c.astype(np.int64), want[20:27])
    st.flush()                                         # nothing to flush: a no-op
    assert st.pending() == (0, 0)
    st.close()
    core.close()


def test_two_stream_objects_on_one_released_plan():
    """The stream object runs the plan through intfft_exec_ws on its own workspace: two objects share one multi-pass plan, also after
    intfft_plan_release_scratch -- and intfft_exec_host on that plan answers INTFFT_ERR_INVALID before touching the device."""
    from intfftk_amd import ERR_INVALID, FrameStream, IntFFTCore, IntFFTError

    core = IntFFTCore(16, 16, 16, 0, 0, "NEW", "FWD")
    assert core.info["scratch_bytes"] > 0
    core.release_scratch()
    x = uniform_frames(24, 1 << 16, 16, 9)
    xin = x.astype(np.int16)
    want = oracle(x, 16, 16, 16, 0, 0, "FWD")
    with pytest.raises(IntFFTError) as ei:
        core.exec_host(xin[:2])
    assert ei.value.status == ERR_INVALID
    a, b = FrameStream(core, 4, 2), FrameStream(core, 3, 3)
    ia = ib = 0
    outa, outb = [], []
    while ia < 24 or ib < 24 or sum(map(len, outa)) < 24 or sum(map(len, outb)) < 24:
        if ia < 24:
            ia += a.push(xin[ia:ia + 5])
            if ia == 24:
                a.flush()
        if ib < 24:
            ib += b.push(xin[ib:ib + 7])
            if ib == 24:
                b.flush()
        ya, yb = a.pull(6), b.pull(6)
        if len(ya):
            outa.append(ya.copy())
        if len(yb):
            outb.append(yb.copy())
    assert np.array_equal(np.concatenate(outa).astype(np.int64), want)
    assert np.array_equal(np.concatenate(outb).astype(np.int64), want)
    a.close()
    b.close()
    core.close()


def test_stream_argument_checks():
    import ctypes

    from intfftk_amd import ERR_INVALID, ERR_NULL, IntFFTCore
    from intfftk_amd import _capi as capi

    L = capi.lib()
    core = IntFFTCore(8, 16, 16, 0, 0, "NEW", "FWD")
    s = ctypes.c_void_p()
    assert L.intfft_stream_open(None, 4, 2, ctypes.byref(s)) == ERR_NULL
    assert L.intfft_stream_open(core._plan, 4, 1, ctypes.byref(s)) == ERR_INVALID
    assert L.intfft_stream_open(core._plan, 4, 65, ctypes.byref(s)) == ERR_INVALID
    assert L.intfft_stream_open(core._plan, 4, 2, ctypes.byref(s)) == 0
    got = ctypes.c_size_t()
    assert L.intfft_stream_push(s, None, 3, None) == ERR_NULL
    assert L.intfft_stream_pull(s, None, 3, ctypes.byref(got), 0) == ERR_NULL
    assert L.intfft_stream_push(s, None, 0, None) == 0
    assert L.intfft_stream_pull(s, None, 0, ctypes.byref(got), 1) == 0 and got.value == 0
    assert L.intfft_stream_close(s) == 0
    assert L.intfft_stream_close(None) == ERR_NULL
    core.close()
