"""PCIe-inclusive rate of the streaming host API (intfft_exec_host) on the C2 workload. Not `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from intfftk_amd import int_fft_single_path

core = int_fft_single_path(10, 16, 16, 0, 0)
rng = np.random.default_rng(1)
x = rng.integers(-2**14, 2**14, size=(65536, 1024, 2), dtype=np.int16)
for chunk in (0, 2048, 8192, 65536):
    core.exec_host(x[:4096], chunk)  # warm
    t0 = time.perf_counter(); y = core.exec_host(x, chunk); dt = time.perf_counter() - t0
    print("chunk_frames=%6d  %.2f ms  %.2f Gsample/s  (%.1f GB/s each way)" % (chunk, dt * 1e3, x.shape[0] * 1024 / dt / 1e9, x.nbytes / dt / 1e9))
# caller-pinned buffers (torch pinned memory): registration is skipped, copies are truly asynchronous
import ctypes, torch
from intfftk_amd import _capi as capi
xp = torch.from_numpy(x).pin_memory(); yp = torch.empty_like(xp).pin_memory()
for chunk in (2048, 8192, 16384):
    capi.check(capi.lib().intfft_exec_host(core._plan, xp.data_ptr(), yp.data_ptr(), 4096, chunk), "warm")
    t0 = time.perf_counter()
    capi.check(capi.lib().intfft_exec_host(core._plan, xp.data_ptr(), yp.data_ptr(), xp.shape[0], chunk), "exec_host")
    dt = time.perf_counter() - t0
    print("pinned chunk_frames=%6d  %.2f ms  %.2f Gsample/s  (%.1f GB/s each way)" % (chunk, dt * 1e3, xp.shape[0] * 1024 / dt / 1e9, x.nbytes / dt / 1e9))
assert np.array_equal(yp.numpy(), y)

# ---- round 6: the frame-queue form (intfft_stream_*): a producer thread pushes bursts, the main thread pulls; PCIe + the two host memcpys
# (caller memory -> pinned ring, pinned ring -> caller memory) are inside the figure
import threading

from intfftk_amd import FrameStream

for slot_frames, n_slots, burst in ((2048, 3, 64), (8192, 3, 512), (8192, 4, 8192)):
    st = FrameStream(core, slot_frames, n_slots)
    out = np.empty_like(x)

    def producer():
        pos = 0
        while pos < x.shape[0]:
            a = st.push(x[pos:pos + burst])
            pos += a
            if a == 0:
                time.sleep(0.00005)
        st.flush()

    for rep in range(2):  # first pass warms the pinned ring's pages
        t0 = time.perf_counter()
        th = threading.Thread(target=producer)
        th.start()
        got = 0
        while got < x.shape[0]:
            yv = st.pull(min(8192, x.shape[0] - got), wait=False, out=out[got:])
            if len(yv):
                got += len(yv)
            else:
                time.sleep(0.00005)
        th.join()
        dt = time.perf_counter() - t0
    assert np.array_equal(out, y)
    print("stream slot_frames=%5d n_slots=%d burst=%5d  %.2f ms  %.2f Gsample/s  (%.1f GB/s each way, host memcpys included)"
          % (slot_frames, n_slots, burst, dt * 1e3, x.shape[0] * 1024 / dt / 1e9, x.nbytes / dt / 1e9))
    st.close()
