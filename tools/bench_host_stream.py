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
